// blsmi.hip -- host side and C ABI (include/blsmi.h) of libblsmi.so.  The kernels live in their own translation units
// (k_pairing_pair.hip: the default lane-pair pairing kernels; k_pairing_single.hip; k_hash.hip; k_curve.hip), declared in
// kernels.h; the verify-path host code is in verify_host.inc.
#include "../../include/blsmi.h"
#include "kernels.h"
#include "lat_programs.h"
#include "util_dev.h"
#include "prepared.h"
#include <functional>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
#include <rccl/rccl.h>                 // types and prototypes only: librccl is dlopen()ed when more than one device is in use
#include <dlfcn.h>
#include <atomic>
#include <thread>
#include <random>

// The level programs of the latency path (gen_lat.py -> lat_programs.bin, 19 MB of 32-byte job descriptors), embedded in the host
// object zlib-compressed (lat_programs.z, 1.3 MB) and inflated once when the library initialises.
#if !defined(__HIP_DEVICE_COMPILE__)
__asm__(".section .rodata\n.balign 256\n.global blsmi_lat_z\n.hidden blsmi_lat_z\nblsmi_lat_z:\n.incbin \"" BLSMI_LAT_BIN "\"\n.global blsmi_lat_z_end\n.hidden blsmi_lat_z_end\nblsmi_lat_z_end:\n.previous\n");
#endif
extern "C" const unsigned char blsmi_lat_z[];
extern "C" const unsigned char blsmi_lat_z_end[];
#include <zlib.h>
static std::vector<unsigned char> g_lat_host;                             // the inflated programs (host copy: program headers are read from it)
#define blsmi_lat_blob (g_lat_host.data())
static int inflate_programs() {                                           // caller holds g_mu
    if (!g_lat_host.empty()) return 0;
    std::vector<unsigned char> out(LAT_TOTAL_BYTES);
    uLongf n = LAT_TOTAL_BYTES;
    if (uncompress(out.data(), &n, blsmi_lat_z, (uLong)(blsmi_lat_z_end - blsmi_lat_z)) != Z_OK || n != LAT_TOTAL_BYTES) return -1;
    g_lat_host.swap(out);
    return 0;
}

namespace {
// Host state.  The library drives a LIST of devices from one process (blsmi_init_devices; blsmi_init binds a single one).
// Every entry point leases one call context (a non-blocking stream, a grow-only scratch buffer and timing events) from
// the pool of one device, so calls from several OS threads (cgo pins one per goroutine call) run concurrently on
// separate HIP streams -- and on separate GPUs when there are several; temporaries come from the device's
// stream-ordered memory pool, whose release threshold is raised so freed blocks are reused by later calls instead of
// returning to the driver.  Large verify batches are split by contiguous block over the devices ("shards"), each
// shard on its own host thread; the only exchanges are the pass/fail bitmap (RCCL all-reduce) and, for one n-way
// VerifyAggregate, the per-device Fq12 partial products (RCCL all-gather) -- DESIGN.md section 5.
std::mutex g_mu;
// One split call at a time owns the per-device exchange buffers (Device::coll); a split call occupies every device anyway.  Lock order:
// g_coll_mu BEFORE g_mu (blsmi_trim only try_locks it while holding g_mu).
std::mutex g_coll_mu;
std::condition_variable g_cv;          // a context was released (lease waiters and shutdown both wait here: notify_all)
bool g_pair_layout = true;              // lane-pair pairing kernels (two lanes per tuple); BLSMI_LAYOUT=single for one tuple per lane
bool g_ready = false;
char g_version[200] = "blsmi 0.7 (uninitialised)";

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "blsmi: %s failed: %s\n", #x, hipGetErrorString(e_)); return BLSMI_E_HIP; } } while (0)

// Grow-only scratch for the Miller-loop -> final-exponentiation hand-off of one call context.
struct Workspace {
    void* p = nullptr; size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
// Temporaries of one call: a bump allocator over grow-only blocks of device memory that belong to the call context.  Every entry
// point used to take its temporaries from the stream-ordered pool (hipMallocAsync / hipFreeAsync); a dozen such pairs cost a
// 2^20-point MSM 2.5 ms of its 14.6 and every small call ~0.1 ms.  A context serves one call at a time and the entry points are
// blocking, so "free" is resetting the offsets when the next call leases the context.  A request goes to the first block with room;
// one that fits nowhere gets a block of its own (at least a quarter of what the arena already holds, so the block count stays
// logarithmic in the peak).  A call that repeats an earlier call's requests finds every one of them in place: steady state makes
// no runtime call at all -- in particular no hipFree, which synchronises the whole device (the first version merged the blocks
// into one on the next lease: a sporadic 20-70 ms inside some later call, seen in bench.py legs).
struct Arena {
    struct Block { char* p; size_t cap, used; };
    std::vector<Block> blocks;
    static constexpr size_t MAX_BLOCKS = 64;
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        for (auto& b : blocks) if (b.used + bytes <= b.cap) { void* r = b.p + b.used; b.used += bytes; return r; }
        size_t total = 0;
        for (auto& b : blocks) total += b.cap;
        const size_t want = std::max(std::max(bytes, (size_t)1 << 20), (total >> 2) & ~(size_t)255);
        void* q = nullptr;
        if (hipMalloc(&q, want) != hipSuccess) {
            (void)hipGetLastError();
            if (want == bytes || hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            blocks.push_back({(char*)q, bytes, bytes});
            return q;
        }
        blocks.push_back({(char*)q, want, bytes});
        return q;
    }
    // mark / rewind: give back what an abandoned attempt allocated since the mark (the MSM's sort buffers, when not all four fit -- the
    // fallback's index buffer may need exactly that memory).  Blocks created after the mark go back to the driver, earlier ones to their fill level.
    struct Mark { std::vector<size_t> used; };
    Mark mark() const { Mark m; for (const auto& b : blocks) m.used.push_back(b.used); return m; }
    void rewind(const Mark& m) {
        while (blocks.size() > m.used.size()) { (void)hipFree(blocks.back().p); blocks.pop_back(); }
        for (size_t i = 0; i < blocks.size(); i++) blocks[i].used = m.used[i];
    }
    void reset() {                                                         // caller: the context's previous call has completed
        if (blocks.size() > MAX_BLOCKS) {                                  // many differently-shaped calls: start over with one block of the total
            size_t total = 0;
            for (auto& b : blocks) total += b.cap;
            release();
            void* q = nullptr;
            if (hipMalloc(&q, total) == hipSuccess) blocks.push_back({(char*)q, total, 0}); else (void)hipGetLastError();
        }
        for (auto& b : blocks) b.used = 0;
    }
    void release() { for (auto& b : blocks) (void)hipFree(b.p); blocks.clear(); }
    size_t total() const { size_t t = 0; for (auto& b : blocks) t += b.cap; return t; }
    // give back blocks, largest first, until at most `keep` bytes remain.  Caller: no work of this context is in flight.  Returns bytes freed.
    size_t trim(size_t keep) {
        size_t t = total(), freed = 0;
        while (t > keep && !blocks.empty()) {
            size_t big = 0;
            for (size_t i = 1; i < blocks.size(); i++) if (blocks[i].cap > blocks[big].cap) big = i;
            (void)hipFree(blocks[big].p);
            t -= blocks[big].cap; freed += blocks[big].cap;
            blocks.erase(blocks.begin() + big);
        }
        return freed;
    }
};
struct Device;
struct Ctx {
    Arena arena;
    hipStream_t stream = nullptr;
    Workspace ws;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    // two side streams + fork/join events: independent kernels of one small call (the two decompressions and the hash of
    // Deserialize + Verify) run side by side instead of one after the other
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    bool busy = false;
    Device* dev = nullptr;
    // what this call put on its device and what the device carried from OTHER calls when it first asked (call_load: the layout choice)
    bool load_set = false; size_t load_n = 0, load_others = 0;
    // per-kernel timing (blsmi_set_profiling): events recorded between the major kernels of the call that holds this context
    std::vector<hipEvent_t> pev; std::vector<const char*> pname; size_t pused = 0;
    bool pclosed = false;               // the call's closing mark is already recorded (on a caller-supplied stream, by UseStream)
    hipError_t ensure_aux() {
        if (aux[0]) return hipSuccess;
        hipError_t e;
        for (auto& a : aux) if ((e = hipStreamCreateWithFlags(&a, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&fork, hipEventDisableTiming)) != hipSuccess) return e;
        for (auto& j : join) if ((e = hipEventCreateWithFlags(&j, hipEventDisableTiming)) != hipSuccess) return e;
        return hipSuccess;
    }
};
constexpr int MAX_CTX = 16;
constexpr int MAX_DEV = 16;
// G1/G2 generators in wire form and the generator's prepared lines, written once per device at init (read-only afterwards)
struct Gens { u8* g1 = nullptr; u8* g2 = nullptr; i32* lines = nullptr; i32* lines_pair = nullptr; u8* lat = nullptr;   // lines_pair: the same lines in the limbs of the lane-pair / lane-quad pairing kernels (14 x 28 bits); lat: the latency-path programs (k_lat.hip)
              i32* fixed1 = nullptr; i32* fixed2 = nullptr; };                                    // fixed-base tables of the generators (k_curve.hip: PrivToPub)
struct Device {
    int id = -1;                        // HIP device ordinal
    int index = 0;                      // position in g_dev
    Ctx ctx[MAX_CTX];
    Gens gens;
    int leased = 0;                     // contexts in use (device choice for unpinned calls)
    std::atomic<size_t> load{0};        // tuples of the pairing / verify calls in flight on this device (call_load)
    unsigned long long leases = 0;      // context leases served since initialisation (blsmi_debug_device_leases)
    // collectives: one RCCL communicator rank and one stream per device, plus a grow-only exchange buffer
    ncclComm_t comm = nullptr;
    hipStream_t coll_stream = nullptr;
    Workspace coll;
};
Device g_dev[MAX_DEV];
int g_ndev = 0;
int g_nshards = 0;                      // logical shards of a split batch (BLSMI_SHARDS; default = number of devices)
size_t g_shard_min = 8192;              // batches below this many tuples are not split (BLSMI_SHARD_MIN)
bool g_force_rccl = false;              // BLSMI_FORCE_RCCL=1: build the communicator even for one device (exercises the collective path on a 1-GPU box)
int g_nctx = 4;                         // BLSMI_STREAMS: contexts per device
int g_rr = 0;                           // round-robin start for unpinned leases
// Retention cap of a call context's temporaries (arena + Miller-loop hand-off buffer): what a context holds beyond this many bytes when
// its call ends goes back to the driver there and then (BLSMI_ARENA_KEEP_MB, default 4096; blsmi_trim for an explicit give-back).
// Without it every context kept the peak of the largest call it ever served -- tens of GB after a burst of 2^20-point calls.
size_t g_arena_keep = (size_t)4096 << 20;
thread_local Ctx* tl_ctx = nullptr;
#define g_stream (tl_ctx->stream)
#define g_ws (tl_ctx->ws)
#define g_gens (tl_ctx->dev->gens)
bool g_use_gen_lines = true;           // BLSMI_GEN_LINES=0 recomputes the generator's lines per tuple (A/B switch)
// optional per-kernel timing (HIP events on the launch stream) for bench.py's roofline object; results are per calling thread
std::atomic<bool> g_profile{false};
thread_local float tl_last_ms[2] = {0.f, 0.f};
// Segments (kernel name, milliseconds) of the profiled calls this thread made since it last read them (blsmi_last_profile).
thread_local std::vector<std::pair<const char*, float>> tl_prof;

// ---- RCCL, loaded on demand (a single-GPU deployment never needs it) ---------------------------------------------
struct Rccl {
    void* h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(const char* path = nullptr) {
        if (h) return true;
        // path: BLSMI_RCCL_PATH (read at initialisation) -- a process without torch (the Go deployment) may have no librccl on its loader
        // path; tried first when given, and its failure is reported rather than papered over by the defaults
        if (path && path[0]) { h = dlopen(path, RTLD_NOW | RTLD_GLOBAL); if (!h) { fprintf(stderr, "blsmi: cannot load BLSMI_RCCL_PATH=%s (%s)\n", path, dlerror()); return false; } }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { if (h) break; h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); }
        if (!h) { fprintf(stderr, "blsmi: cannot load librccl (%s): multi-GPU sharding needs RCCL (BLSMI_RCCL_PATH names a copy)\n", dlerror()); return false; }
#define BLSMI_SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(h, "nccl" #f)); if (!f) { fprintf(stderr, "blsmi: librccl lacks nccl" #f "\n"); return false; }
        BLSMI_SYM(CommInitAll) BLSMI_SYM(CommDestroy) BLSMI_SYM(AllReduce) BLSMI_SYM(AllGather) BLSMI_SYM(GroupStart) BLSMI_SYM(GroupEnd) BLSMI_SYM(GetErrorString)
#undef BLSMI_SYM
        return true;
    }
} g_rccl;
bool g_have_comm = false;
// TEST HOOK (BLSMI_DEVICE_ALIAS=0,0[,0,0], read by blsmi_init_devices): N LOGICAL devices -- each with its own context pool, streams,
// generator tables and exchange buffer, exactly what a device of an N-GPU node gets -- that all sit on the physical GPUs the list names.
// RCCL refuses two ranks on one GPU, so under the hook (and only there) the two collectives are a host-staged stand-in with the same
// semantics (coll_* in verify_host.inc); the product path is RCCL.  A one-GPU box thereby executes g_dev[d] for d > 0, the
// `shard % ndev` routing, the per-device pools and the owner routing of the *_dev entry points (blsmi_debug_alias_own).
bool g_alias = false;
struct AliasRange { const char* lo; const char* hi; int dev; };
std::vector<AliasRange> g_alias_own;   // device-pointer ranges assigned to a logical device (g_mu)
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "blsmi: %s failed: %s\n", #x, g_rccl.GetErrorString(r_)); return BLSMI_E_RCCL; } } while (0)

inline unsigned nblocks(size_t n) { return (unsigned)((n + WG - 1) / WG); }
int init_device(Device& d) {            // caller holds g_mu
    HIPCHK(hipSetDevice(d.id));
    hipMemPool_t pool;
    HIPCHK(hipDeviceGetDefaultMemPool(&pool, d.id));
    uint64_t keep = UINT64_MAX;
    HIPCHK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    for (auto& c : d.ctx) c.dev = &d;
    // The call contexts' streams, created back to back before any other stream of the library: the HIP runtime hands its hardware queues
    // (GPU_MAX_HW_QUEUES, default 4) to streams in creation order, and kernels of one queue run one after the other -- created lazily, between
    // side streams and the exchange stream, two of four contexts shared a queue (rocprofv3 --kernel-trace of four concurrent callers: three queues)
    for (int i = 0; i < g_nctx; i++) if (!d.ctx[i].stream) HIPCHK(hipStreamCreateWithFlags(&d.ctx[i].stream, hipStreamNonBlocking));
    if (!d.gens.g1) {
        HIPCHK(hipMalloc((void**)&d.gens.g1, 96)); HIPCHK(hipMalloc((void**)&d.gens.g2, 192));
        hipLaunchKernelGGL(k_write_generators, dim3(1), dim3(WG), 0, nullptr, d.gens.g1, d.gens.g2);
        HIPCHK(hipMalloc((void**)&d.gens.lines, sizeof(i32) * 68 * 3 * 2 * NL));
        hipLaunchKernelGGL(k_prepare_generator_lines, dim3(1), dim3(WG), 0, nullptr, (const u8*)d.gens.g2, d.gens.lines);
        HIPCHK(hipMalloc((void**)&d.gens.lines_pair, sizeof(i32) * 68 * 3 * 2 * NL));
        hipLaunchKernelGGL(k_prepare_generator_lines_pair, dim3(1), dim3(WG), 0, nullptr, (const u8*)d.gens.g2, d.gens.lines_pair);
        HIPCHK(hipGetLastError());
        if (inflate_programs()) { fprintf(stderr, "blsmi: the embedded level programs do not inflate\n"); return BLSMI_E_ARG; }
        HIPCHK(hipMalloc((void**)&d.gens.lat, LAT_TOTAL_BYTES));
        HIPCHK(hipMemcpy(d.gens.lat, blsmi_lat_blob, LAT_TOTAL_BYTES, hipMemcpyHostToDevice));
        // fixed-base tables of the generators (PrivToPub): [d 256^w] G for d = 1 .. 255, w = 0 .. 31, computed by the
        // scalar-multiplication kernels themselves (the generators lie in the subgroup: endomorphism ladder)
        {
            constexpr size_t NE = 32 * 255;
            std::vector<uint8_t> sc(NE * 32, 0);
            for (int w = 0; w < 32; w++) for (int dd = 1; dd < 256; dd++) sc[((size_t)w * 255 + dd - 1) * 32 + 31 - w] = (uint8_t)dd;
            u8 *dsc = nullptr, *wire = nullptr, *inf = nullptr;
            HIPCHK(hipMalloc((void**)&dsc, NE * 32)); HIPCHK(hipMalloc((void**)&wire, NE * 192)); HIPCHK(hipMalloc((void**)&inf, NE));
            HIPCHK(hipMemcpy(dsc, sc.data(), NE * 32, hipMemcpyHostToDevice));
            HIPCHK(hipMalloc((void**)&d.gens.fixed1, sizeof(i32) * NE * 2 * NL)); HIPCHK(hipMalloc((void**)&d.gens.fixed2, sizeof(i32) * NE * 4 * NL));
            hipLaunchKernelGGL(k_g1_mul_glv, dim3(nblocks(NE)), dim3(WG), 0, nullptr, (const u8*)d.gens.g1, (size_t)0, (const u8*)dsc, wire, inf, NE);
            hipLaunchKernelGGL(k_fixed_table_from_wire, dim3(nblocks(NE * 2)), dim3(WG), 0, nullptr, (const u8*)wire, 2, d.gens.fixed1, NE);
            hipLaunchKernelGGL(k_g2_mul_glv_pair, dim3((unsigned)((NE + PT - 1) / PT)), dim3(WG), 0, nullptr, (const u8*)d.gens.g2, (size_t)0, (const u8*)dsc, wire, inf, NE);
            hipLaunchKernelGGL(k_fixed_table_from_wire, dim3(nblocks(NE * 4)), dim3(WG), 0, nullptr, (const u8*)wire, 4, d.gens.fixed2, NE);
            HIPCHK(hipGetLastError());
            HIPCHK(hipDeviceSynchronize());
            (void)hipFree(dsc); (void)hipFree(wire); (void)hipFree(inf);
        }
        HIPCHK(hipDeviceSynchronize());
    }
    return BLSMI_OK;
}
// Latency path (k_lat.hip): batches of at most g_lat_max tuples run one tuple per WAVE instead of one per lane pair.
// (2 048 waves fit the chip at once; beyond a few thousand tuples the lane-pair kernels win on throughput.)
// per-call opt-out of the endomorphism ladders (the *_ex entry points with BLSMI_MUL_ANY_POINT): set on every thread that works for such a call
thread_local bool tl_mul_any = false;
struct MulAny { bool saved; explicit MulAny(bool on) : saved(tl_mul_any) { tl_mul_any = on; } ~MulAny() { tl_mul_any = saved; } };
std::atomic<bool> g_mul_subgroup{true}; // scalar multiplication through the endomorphisms (multiplicands in the subgroup); BLSMI_MUL_GENERIC=1 / blsmi_set_mul_assume_subgroup(0): plain ladder
std::atomic<size_t> g_lat_max{8192};    // BLSMI_LAT_MAX, blsmi_set_latency_threshold (read by every call, written rarely).  The two paths meet at ~10 000 tuples
                                        // for pairings and verifies alike (tools/crossover.py: 8192 pairings 8.4 ms against 10.7, 16 384: 16.4 against 11.3)
// Three layouts by batch size (pairings and verifies alike; tools/midsize.py): one tuple per WAVE up to min(g_lat_max, g_quad_min) tuples,
// one per lane QUAD up to g_quad_max (16 384 tuples = one wave on every SIMD), one per lane PAIR beyond (65 536 fill the chip twice over).
std::atomic<size_t> g_quad_max{16384};  // BLSMI_QUAD_MAX, blsmi_set_quad_threshold (0: no quad kernels)
std::atomic<size_t> g_quad_min{5632};   // BLSMI_QUAD_MIN: the quad kernels take over from the latency path here already (pairings: 5.9 ms flat against 1 ms per 1 024 tuples; verifies 8.7 against 1.5)
// The thresholds above are a LONE caller's: one tuple per wave finishes 4 096 pairings in 4.3 ms where the quad kernels take their flat 6 ms.
// Callers that arrive together are a different matter (tools/midsize_concurrency.py): the latency path saturates the chip at 1.04 M pairings/s
// whatever the number of calls in flight (its waves are bounded by LDS, 9 per CU), the quad kernels at 2.8 M/s (two 8 192-tuple calls take
// the 6 ms of one).  So the choice goes by what the DEVICE carries: a call of at least g_crowd_floor tuples takes the quad kernels when its
// tuples plus those of the other calls in flight pass the lone crossover (floor 1 536: four callers x 2 048 pairings 7.9 -> 6.1 ms a call; at 1 024
// the sum never passes the crossover with four contexts).  A call registers its tuples at its first layout question and
// keeps the answer's input for its whole life (one call never sees two different loads); ~CtxLease takes them off again.
std::atomic<bool> g_crowd_quad{true};        // BLSMI_CROWD_QUAD / blsmi_set_option("crowd_quad")
std::atomic<size_t> g_crowd_floor{1536};     // BLSMI_CROWD_FLOOR / blsmi_set_option("crowd_floor")
std::atomic<size_t> g_assume_load{0};        // blsmi_set_option("assume_load"): test hook, tuples pretended to be in flight from other calls
inline size_t call_load(size_t n) {
    Ctx* c = tl_ctx;
    if (!c || !c->dev) return 0;
    if (!c->load_set) { c->load_others = c->dev->load.fetch_add(n, std::memory_order_relaxed); c->load_n = n; c->load_set = true; }
    return c->load_others + g_assume_load.load(std::memory_order_relaxed);
}
// A fourth layout between the wave and the quad (round 6, k_pairing_row.hip): one tuple per DPP ROW of sixteen lanes.  g_row_min .. g_row_max
// tuples (2 048 .. 8 192) of a LONE caller take it: 4 096 tuples are one wave on every SIMD there (a quarter of the SIMDs in the quad layout, four waves of
// 3.3 x the instructions on the one-tuple-per-wave path).  When other calls are in flight on the device the choice above stands (the quad
// kernels spend fewer lane-instructions per tuple: 12.6 M against 18 M).  blsmi_set_row_threshold / BLSMI_ROW_MIN / BLSMI_ROW_MAX; max 0: off.
std::atomic<size_t> g_hash_row_min{2048}, g_hash_row_max{4096};   // blsmi_set_option("hash_row_min" / "hash_row_max"): HashG2 of this many messages clears its cofactor in the lane-row layout (k_clear_h2_row; max 0: never)
std::atomic<size_t> g_hash_quad_min{4097}, g_hash_quad_max{16384};   // "hash_quad_min" / "hash_quad_max": ... four lanes per message (k_clear_h2_quad)
std::atomic<size_t> g_hash_oct_min{2048}, g_hash_oct_max{7168};   // "hash_oct_min" / "hash_oct_max": ... eight lanes per message (k_clear_h2_oct, oct_g2.inc); takes precedence over the row and quad tails
std::atomic<size_t> g_hash_g1_quad_min{1280}, g_hash_g1_quad_max{32768};   // "hash_g1_quad_min" / "hash_g1_quad_max": HashG1's tail four lanes per message (k_hash_g1_finish_quad)
std::atomic<size_t> g_swu_row_max{4096};   // "swu_row_max": the SWU maps of HashG1 / HashG2 run a ROW of sixteen lanes per map (k_swu_g?_rows) above BLSMI_SWU_WAVE_MAX and up to here; 0: never
std::atomic<size_t> g_row_side_piece{0};   // "row_side_piece": the side stream runs its row kernel in launches of this many tuples (0, the default: one launch -- pieces measured slower, verify_host.inc)
std::atomic<size_t> g_row_side_lds{0};   // "row_side_lds": bytes of (unused) LDS per workgroup of the side kernel of a call of more than 4 096 tuples -- 40960: four workgroups a CU, a wave slot of every SIMD left to the hash; measured slower (verify_host.inc); 0, the default: none
std::atomic<bool> g_row_side_g2pubs{true};   // "row_side_g2pubs": ... for g2pubs too (the signature side over the generator's table as a kernel of its own; measured, see verify_host.inc)
std::atomic<bool> g_row_side{true};     // BLSMI_ROW_SIDE / blsmi_set_option("row_side"): a Verify in the row layout runs its signature side beside the hash (verify_host.inc)
std::atomic<size_t> g_row_min{2048};   // (tools/midsize4.py: 2 048 pairings 2.08 against 2.21 ms on the wave path, g1pubs verifies 4.45 against 4.72, g2pubs 3.66 against 3.57; 1 024: 2.05 against 1.54)
std::atomic<size_t> g_row_max{8192};   // (8 192 pairings 4.0 ms against the quad kernels' flat 5.7; 12 288: 6+ against 5.7)
// pairing_only: blsmi_pairing_batch has no hash beside its two kernels, and there the row kernels stay ahead of the quad kernels' flat 5.7 ms up to 12 288 tuples
// (three waves per SIMD: 5.34 ms; verifies cross at ~10 000: 12 288 g2pubs verifies 8.96 against 7.95 ms) -- half again the general maximum
inline bool use_row(size_t n, bool pairing_only = false) {
    const bool crowd = g_crowd_quad.load(std::memory_order_relaxed) && n >= g_crowd_floor.load(std::memory_order_relaxed);
    const size_t others = crowd ? call_load(n) : 0;                        // (every sizeable call is counted, whatever layout it takes itself)
    size_t hi = g_row_max.load(std::memory_order_relaxed);
    if (pairing_only) hi += hi / 2;
    if (!g_pair_layout || hi == 0 || n > hi || n < g_row_min.load(std::memory_order_relaxed)) return false;
    const size_t lone = std::min(g_lat_max.load(), g_quad_min.load());
    return !(others > 0 && n + others > lone);
}
inline bool use_quad(size_t n) {
    // every sizeable call is counted, whatever layout it takes itself (a 65 536-tuple call in flight is load for the 3 000-tuple call beside it)
    const bool crowd = g_crowd_quad.load(std::memory_order_relaxed) && n >= g_crowd_floor.load(std::memory_order_relaxed);
    const size_t others = crowd ? call_load(n) : 0;
    if (!g_pair_layout || n > g_quad_max) return false;
    if (use_row(n)) return false;
    const size_t lone = std::min(g_lat_max.load(), g_quad_min.load());
    return n > lone || (crowd && n + others > lone);
}
inline bool use_lat(size_t n) { return n <= g_lat_max && !use_quad(n) && !use_row(n); }
inline unsigned qblocks(size_t n) { return (unsigned)((n + QT - 1) / QT); }
inline unsigned rblocks(size_t n) { return (unsigned)((n + RT - 1) / RT); }
inline bool mul_subgroup() { return !tl_mul_any && g_mul_subgroup.load(std::memory_order_relaxed); }
inline u32 lat_lds_bytes(size_t prog_offset) { u32 nslot; memcpy(&nslot, blsmi_lat_blob + prog_offset + 8, 4); return nslot * 80; }   // k_lat.hip: SLOT_WORDS * 4
// ---- the environment is read ONCE, at initialisation (under g_mu), never from an entry point ---------------------------------------
// getenv racing a setenv in another thread is undefined (glibc), and a variable read per call silently changes what a concurrent caller
// gets.  Every BLSMI_* variable is therefore read here and nowhere else; what the A/B tests flip at run time are the three
// atomics below, through blsmi_set_option.  The variables are listed in include/blsmi.h ("Environment").
struct EnvCfg {
    size_t swu_wave_max = 512;          // BLSMI_SWU_WAVE_MAX: the smallest hashes / decompressions run one WAVE per field exponentiation up to here (measured: 0.68 against 0.96 ms at 512 messages, 1.18 against 0.98 at 1 024)
    bool hash_g1_split = true;          // BLSMI_HASH_G1_SPLIT
    bool hash_g2_pair = true;           // BLSMI_HASH_G2_PAIR=0 keeps the one-lane HashG2 kernel
    unsigned hash_g2_pair_redo_every = 0;   // BLSMI_HASH_G2_PAIR_REDO_EVERY (tests: exercise the redo pass)
    bool cofac2_pair = true;            // BLSMI_COFAC2_PAIR=0 keeps the fused one-lane HashG2WithDomain kernel
    long long sig_side_max = -1;        // BLSMI_SIG_SIDE_MAX (-1: the per-package defaults)
    size_t side_max = 131072;           // BLSMI_SIDE_MAX
    size_t fixed_wave_max = 2048;       // BLSMI_FIXED_WAVE_MAX
    size_t msm_bucket_min = (size_t)1 << 17;   // BLSMI_MSM_BUCKET_MIN
    size_t combine_max = 1024; int combine_wait_us = 150, combine_inflight_max = 2; bool combine_debug = false;   // BLSMI_COMBINE_*
    char rccl_path[512] = {0};          // BLSMI_RCCL_PATH: librccl for a process that has none mapped yet (a Go binary has no torch to bring it)
} g_env;
std::atomic<bool> g_agg_cofactor_pow{true};    // BLSMI_AGG_COFACTOR_POW / blsmi_set_option("agg_cofactor_pow")
std::atomic<bool> g_msm_sort{true};            // BLSMI_MSM_SORT / blsmi_set_option("msm_sort")
std::atomic<bool> g_lat_rolled{true};          // BLSMI_LAT_ROLLED / blsmi_set_option("lat_rolled"): 0 = small Pairing calls take the STRAIGHT-LINE copy of their level program (pairing1s) instead of the one with rolled squaring runs (A/B)
std::atomic<size_t> g_combine_mid_max{8192};  // BLSMI_COMBINE_MID_MAX / blsmi_set_option("combine_mid_max"): concurrent Verify calls of BLSMI_COMBINE_MAX <= n < this many tuples merge into one launch (verify_host.inc); 0: never
std::atomic<bool> g_dup_force_sort{false};     // BLSMI_DUP_FORCE_SORT / blsmi_set_option("dup_force_sort"): test hook, the duplicate screen's fallback on every call
// Options a caller has set through the API (blsmi_set_option, blsmi_set_*_threshold, blsmi_set_mul_assume_subgroup) keep their values when the library
// (re-)initialises: the environment and the defaults apply only to what was never set explicitly (a set_option before the first entry point,
// or before a re-initialisation after blsmi_shutdown, was silently overwritten here).
enum { X_AGG_POW, X_MSM_SORT, X_DUP_SORT, X_LAT_ROLLED, X_CROWD_QUAD, X_COMBINE_MID, X_CROWD_FLOOR, X_ROW_SIDE, X_LAT_MAX, X_QUAD_MAX, X_ROW, X_MUL_SUBGROUP };
std::atomic<unsigned> g_explicit{0};
inline void set_explicit(int bit) { g_explicit.fetch_or(1u << bit, std::memory_order_relaxed); }
inline bool is_explicit(int bit) { return (g_explicit.load(std::memory_order_relaxed) >> bit) & 1u; }
void load_env() {                       // caller holds g_mu; runs once per initialisation
    auto num = [](const char* name, size_t dflt) { const char* v = getenv(name); return v ? (size_t)strtoull(v, nullptr, 10) : dflt; };
    auto flag = [](const char* name, bool dflt) { const char* v = getenv(name); return v ? atoi(v) != 0 : dflt; };
    g_env.swu_wave_max = num("BLSMI_SWU_WAVE_MAX", 512);
    g_env.hash_g1_split = flag("BLSMI_HASH_G1_SPLIT", true);
    g_env.hash_g2_pair = flag("BLSMI_HASH_G2_PAIR", true);
    g_env.hash_g2_pair_redo_every = (unsigned)num("BLSMI_HASH_G2_PAIR_REDO_EVERY", 0);
    g_env.cofac2_pair = flag("BLSMI_COFAC2_PAIR", true);
    { const char* v = getenv("BLSMI_SIG_SIDE_MAX"); g_env.sig_side_max = v ? atoll(v) : -1LL; }
    g_env.side_max = num("BLSMI_SIDE_MAX", 131072);
    g_env.fixed_wave_max = num("BLSMI_FIXED_WAVE_MAX", 2048);
    g_env.msm_bucket_min = num("BLSMI_MSM_BUCKET_MIN", (size_t)1 << 17);
    g_env.combine_max = num("BLSMI_COMBINE_MAX", 1024);
    if (!is_explicit(X_COMBINE_MID)) g_combine_mid_max = num("BLSMI_COMBINE_MID_MAX", 8192);
    g_env.combine_wait_us = (int)num("BLSMI_COMBINE_WAIT_US", 150);
    g_env.combine_inflight_max = std::max(1, (int)num("BLSMI_COMBINE_INFLIGHT", 2));
    g_env.combine_debug = getenv("BLSMI_COMBINE_DEBUG") != nullptr;
    { const char* v = getenv("BLSMI_RCCL_PATH"); snprintf(g_env.rccl_path, sizeof g_env.rccl_path, "%s", v ? v : ""); }
    if (!is_explicit(X_AGG_POW)) { const char* v = getenv("BLSMI_AGG_COFACTOR_POW"); g_agg_cofactor_pow = !(v && v[0] == '0'); }
    if (!is_explicit(X_MSM_SORT)) { const char* v = getenv("BLSMI_MSM_SORT"); g_msm_sort = !(v && v[0] == '0'); }
    if (!is_explicit(X_DUP_SORT)) g_dup_force_sort = getenv("BLSMI_DUP_FORCE_SORT") != nullptr;
    if (!is_explicit(X_LAT_ROLLED)) { const char* v = getenv("BLSMI_LAT_ROLLED"); g_lat_rolled = !(v && v[0] == '0'); }
    if (!is_explicit(X_CROWD_QUAD)) { const char* v = getenv("BLSMI_CROWD_QUAD"); g_crowd_quad = !(v && v[0] == '0'); }
    if (!is_explicit(X_CROWD_FLOOR)) if (const char* v = getenv("BLSMI_CROWD_FLOOR")) g_crowd_floor = (size_t)strtoull(v, nullptr, 10);
}
// devs[0..ndev): HIP ordinals.  Caller holds g_mu.
int ensure_init_list(const int* devs, int ndev) {
    if (g_ready) return BLSMI_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return BLSMI_E_NODEVICE;
    if (ndev < 1 || ndev > MAX_DEV) return BLSMI_E_ARG;
    for (int i = 0; i < ndev; i++) {
        if (devs[i] < 0 || devs[i] >= count) return BLSMI_E_ARG;
        for (int j = 0; j < i; j++) if (devs[j] == devs[i] && !g_alias) return BLSMI_E_ARG;
    }
    load_env();
    const char* lay = getenv("BLSMI_LAYOUT");
    g_pair_layout = !(lay && std::string(lay) == "single");      // default: lane-pair kernels; BLSMI_LAYOUT=single selects one tuple per lane
    if (const char* ns = getenv("BLSMI_STREAMS")) { int v = atoi(ns); g_nctx = v < 1 ? 1 : (v > MAX_CTX ? MAX_CTX : v); }
    const char* gl = getenv("BLSMI_GEN_LINES");
    g_use_gen_lines = !(gl && std::string(gl) == "0");
    if (!is_explicit(X_LAT_MAX)) if (const char* v = getenv("BLSMI_LAT_MAX")) g_lat_max = (size_t)strtoull(v, nullptr, 10);
    if (!is_explicit(X_QUAD_MAX)) if (const char* v = getenv("BLSMI_QUAD_MAX")) g_quad_max = (size_t)strtoull(v, nullptr, 10);
    if (const char* v = getenv("BLSMI_QUAD_MIN")) g_quad_min = (size_t)strtoull(v, nullptr, 10);
    if (!is_explicit(X_ROW)) {
        if (const char* v = getenv("BLSMI_ROW_MIN")) g_row_min = (size_t)strtoull(v, nullptr, 10);
        if (const char* v = getenv("BLSMI_ROW_MAX")) g_row_max = (size_t)strtoull(v, nullptr, 10);
    }
    if (!is_explicit(X_ROW_SIDE)) { const char* v = getenv("BLSMI_ROW_SIDE"); g_row_side = !(v && v[0] == '0'); }
    if (const char* v = getenv("BLSMI_ARENA_KEEP_MB")) g_arena_keep = (size_t)strtoull(v, nullptr, 10) << 20;
    if (!is_explicit(X_MUL_SUBGROUP)) if (const char* v = getenv("BLSMI_MUL_GENERIC")) g_mul_subgroup = std::string(v) == "0";
    g_force_rccl = getenv("BLSMI_FORCE_RCCL") != nullptr && std::string(getenv("BLSMI_FORCE_RCCL")) != "0";
    for (int i = 0; i < ndev; i++) {
        g_dev[i].id = devs[i]; g_dev[i].index = i; g_dev[i].leases = 0;
        int rc = init_device(g_dev[i]);
        if (rc) return rc;
    }
    g_ndev = ndev;
    g_nshards = ndev;
    if (const char* v = getenv("BLSMI_SHARDS")) { int k = atoi(v); if (k >= 1 && k <= 64) g_nshards = k; }
    if (const char* v = getenv("BLSMI_SHARD_MIN")) g_shard_min = (size_t)strtoull(v, nullptr, 10);
    if (g_alias) {                                                         // host-staged stand-in for the collectives (test hook)
        for (int i = 0; i < ndev; i++) { HIPCHK(hipSetDevice(g_dev[i].id)); HIPCHK(hipStreamCreateWithFlags(&g_dev[i].coll_stream, hipStreamNonBlocking)); }
        g_have_comm = ndev > 1;
    } else if (ndev > 1 || g_force_rccl) {
        if (!g_rccl.load(g_env.rccl_path)) return BLSMI_E_RCCL;
        ncclComm_t comms[MAX_DEV];
        NCCLCHK(g_rccl.CommInitAll(comms, ndev, devs));
        for (int i = 0; i < ndev; i++) {
            g_dev[i].comm = comms[i];
            HIPCHK(hipSetDevice(g_dev[i].id));
            HIPCHK(hipStreamCreateWithFlags(&g_dev[i].coll_stream, hipStreamNonBlocking));
        }
        g_have_comm = true;
    }
    HIPCHK(hipSetDevice(g_dev[0].id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, g_dev[0].id));
    snprintf(g_version, sizeof g_version, "blsmi 0.7 %s CUs=%d devices=%d shards=%d%s", prop.gcnArchName, prop.multiProcessorCount, g_ndev, g_nshards, g_alias ? " ALIASED-DEVICES(test hook: host-staged collectives)" : g_have_comm ? " rccl" : "");
    g_ready = true;
    return BLSMI_OK;
}
int ensure_init_default() {             // lazy initialisation by the first entry point: device 0 only (blsmi_init_devices opts into more)
    const int d0 = 0;
    return ensure_init_list(&d0, 1);
}
int device_index_of_ordinal(int ordinal) { for (int i = 0; i < g_ndev; i++) if (g_dev[i].id == ordinal) return i; return -1; }

// prof_mark(name): called right before the launch of a major kernel (name = nullptr: after the last one).  Segment i runs from
// mark i to mark i + 1 on the launch stream and carries the name of mark i; the marks of a call are turned into
// (name, ms) pairs when its context lease ends (every entry point is blocking, so the stream is idle by then).
#define g_stream_ (tl_ctx->stream)
inline void prof_mark(const char* name) {
    if (!g_profile.load(std::memory_order_relaxed) || !tl_ctx) return;
    Ctx& c = *tl_ctx;
    if (c.pused == c.pev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c.pev.push_back(e); c.pname.push_back(nullptr); }
    if (hipEventRecord(c.pev[c.pused], g_stream_) != hipSuccess) return;
    c.pname[c.pused++] = name;
}
inline void prof_collect(Ctx& c) {
    if (c.pused < 2) { c.pused = 0; return; }
    (void)hipEventSynchronize(c.pev[c.pused - 1]);
    for (size_t i = 0; i + 1 < c.pused; i++) {
        float ms = 0.f;
        // an unnamed segment is the time between the end of one marked kernel and the start of the next: small kernels, copies,
        // allocation and host work the stream waited for -- reported as "(between)" so that the marks add up to the call
        if (hipEventElapsedTime(&ms, c.pev[i], c.pev[i + 1]) == hipSuccess) tl_prof.emplace_back(c.pname[i] ? c.pname[i] : "(between)", ms);
    }
    if (tl_prof.size() > 16384) tl_prof.erase(tl_prof.begin(), tl_prof.begin() + 8192);   // nobody is reading: keep the newest
    c.pused = 0;
}

// Lease of one call context for the duration of an entry point.  dev_index < 0: any device (the one with the fewest
// contexts in use, scanning from a rotating start, so that concurrent callers spread over the GPUs).
struct CtxLease {
    int rc = BLSMI_OK;
    Ctx* mine = nullptr;
    Ctx* outer = nullptr;                // a lease taken while this thread already holds one (nested entry points) is restored on release
    explicit CtxLease(int dev_index = -1) {
        std::unique_lock<std::mutex> lk(g_mu);
        rc = ensure_init_default();
        if (rc) return;
        if (dev_index >= g_ndev) { rc = BLSMI_E_ARG; return; }
        Ctx* c = nullptr;
        for (;;) {
            int best = -1;
            for (int k = 0; k < g_ndev; k++) {
                const int i = dev_index >= 0 ? dev_index : (g_rr + k) % g_ndev;
                if (g_dev[i].leased < g_nctx && (best < 0 || g_dev[i].leased < g_dev[best].leased)) best = i;
                if (dev_index >= 0) break;
            }
            if (best >= 0) { for (int i = 0; i < g_nctx && !c; i++) if (!g_dev[best].ctx[i].busy) c = &g_dev[best].ctx[i]; }
            if (c) break;
            g_cv.wait(lk);
        }
        if (dev_index < 0) g_rr = (g_rr + 1) % g_ndev;
        if (hipSetDevice(c->dev->id) != hipSuccess) { rc = BLSMI_E_HIP; return; }      // the current device is per-thread state
        if (!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = BLSMI_E_HIP; return; }
        c->busy = true; c->dev->leased++; c->dev->leases++;
        c->pused = 0; c->pclosed = false;
        c->load_set = false; c->load_n = 0; c->load_others = 0;
        c->arena.reset();                                                   // the temporaries of the context's previous call (its streams are idle: see the destructor)
        outer = tl_ctx;
        tl_ctx = mine = c;
    }
    ~CtxLease() {
        if (!mine) return;
        if (mine->pused) { if (!mine->pclosed) prof_mark(nullptr); prof_collect(*mine); }
        // Entry points are blocking, so on every normal return the context's streams are idle and these queries cost a microsecond.
        // An ERROR return (HIPCHK, a nonzero rc) leaves without synchronising: drain whatever it had enqueued, so that the next call on
        // this context does not recycle arena memory under a kernel or copy that is still running.
        if (mine->stream && hipStreamQuery(mine->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(mine->stream); }
        for (auto a : mine->aux) if (a && hipStreamQuery(a) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(a); }
        if (mine->arena.total() + mine->ws.cap > g_arena_keep) {           // retention cap: an outsized call's temporaries go back now
            if (mine->ws.cap > g_arena_keep / 2) mine->ws.release();
            (void)mine->arena.trim(g_arena_keep > mine->ws.cap ? g_arena_keep - mine->ws.cap : 0);
        }
        if (mine->load_set) { mine->dev->load.fetch_sub(mine->load_n, std::memory_order_relaxed); mine->load_set = false; }
        { std::lock_guard<std::mutex> lk(g_mu); mine->busy = false; mine->dev->leased--; }
        tl_ctx = outer;
        if (outer) (void)hipSetDevice(outer->dev->id);
        g_cv.notify_all();
    }
};

// is a call context free on some device right now?  (a hint for the request combiner: the answer may be stale by the time it is used)
bool context_free_now() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ready) return true;
    for (int i = 0; i < g_ndev; i++) if (g_dev[i].leased < g_nctx) return true;
    return false;
}
// A caller-supplied stream (the *_dev entry points) stands in for the leased context's stream for one call.
struct UseStream {
    hipStream_t saved;
    bool mine;
    explicit UseStream(void* s) : saved(tl_ctx->stream), mine(s != nullptr) { if (s) tl_ctx->stream = (hipStream_t)s; }
    ~UseStream() {
        if (mine) {
            // the closing profile mark belongs on the stream the kernels ran on, not on the context's own stream
            if (tl_ctx->pused && !tl_ctx->pclosed) { prof_mark(nullptr); tl_ctx->pclosed = true; }
            // an error return leaves work on the CALLER's stream: drain it before the call's temporaries can be recycled
            if (hipStreamQuery(tl_ctx->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(tl_ctx->stream); }
        }
        tl_ctx->stream = saved;
    }
};
// device (index into g_dev) that owns a device pointer handed to a *_dev entry point; -1 if it is not one of ours
int device_index_of_pointer(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_alias) for (auto& r : g_alias_own) if ((const char*)p >= r.lo && (const char*)p < r.hi) return r.dev < g_ndev ? r.dev : -1;
    return device_index_of_ordinal(a.device);
}

// Device temporary of the current call, from the leased context's arena (released when the context is leased again)
struct DBuf {
    void* p = nullptr;
    hipError_t alloc(size_t bytes, hipStream_t = nullptr) { p = tl_ctx->arena.alloc(bytes ? bytes : 1); return p ? hipSuccess : hipErrorOutOfMemory; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- sharding ---------------------------------------------------------------------------------------------------
// A batch of n independent tuples is cut into g_nshards contiguous blocks (block boundaries on multiples of `align`
// tuples); shard k runs on device k mod g_ndev, on its own host thread (the current device is per-thread state), under
// its own context lease.  body(shard, lo, hi) returns a BLSMI_* code; the first failure is reported.
struct ShardPlan { int nshards = 1; size_t n = 0, per = 0; size_t lo(int k) const { return std::min(n, per * (size_t)k); } size_t hi(int k) const { return std::min(n, per * (size_t)(k + 1)); } };
ShardPlan plan_shards(size_t n, size_t align) {
    ShardPlan p; p.n = n;
    int k = g_nshards;
    if (n < g_shard_min || k <= 1) k = 1;
    size_t per = (n + k - 1) / k;
    per = (per + align - 1) / align * align;
    p.nshards = k; p.per = per ? per : align;
    return p;
}
template <class F>
int run_shards(const ShardPlan& plan, F&& body) {
    if (plan.nshards == 1) { CtxLease lease; if (lease.rc) return lease.rc; return body(0, plan.lo(0), plan.hi(0)); }
    std::vector<int> rcs(plan.nshards, BLSMI_OK);
    std::vector<std::thread> th;
    auto run = [&](int k) {
        if (plan.lo(k) >= plan.hi(k)) return;
        CtxLease lease(k % g_ndev);
        rcs[k] = lease.rc ? lease.rc : body(k, plan.lo(k), plan.hi(k));
    };
    for (int k = 1; k < plan.nshards; k++) th.emplace_back(run, k);
    run(0);
    for (auto& t : th) t.join();
    for (int rc : rcs) if (rc) return rc;
    return BLSMI_OK;
}
}  // namespace
#define LOCK_AND_INIT() CtxLease lease_; if (lease_.rc) return lease_.rc;
// *_dev entry points run on the device that owns the caller's buffers
#define LOCK_AND_INIT_AT(ptr) const int devidx_ = device_index_of_pointer_init(ptr); if (devidx_ < 0) return devidx_ == -1 ? BLSMI_E_ARG : devidx_; CtxLease lease_(devidx_); if (lease_.rc) return lease_.rc;

#define BLSMI_API extern "C" __attribute__((visibility("default")))

namespace {
int device_index_of_pointer_init(const void* p) {                          // -1: not a pointer on one of the library's devices; < -1: init failure
    { std::lock_guard<std::mutex> lk(g_mu); int rc = ensure_init_default(); if (rc) return rc < -1 ? rc : -2; }
    return device_index_of_pointer(p);
}
}  // namespace

BLSMI_API int blsmi_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ready) return device_index_of_ordinal(device) >= 0 ? BLSMI_OK : BLSMI_E_ARG;
    return ensure_init_list(&device, 1);
}
BLSMI_API int blsmi_init_devices(int ndev) {
    std::lock_guard<std::mutex> lk(g_mu);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return BLSMI_E_NODEVICE;
    int devs[MAX_DEV];
    int nalias = 0;
    if (const char* al = getenv("BLSMI_DEVICE_ALIAS")) {                   // test hook: logical devices on the listed physical ones
        for (const char* q = al; *q && nalias < MAX_DEV;) {
            char* end = nullptr;
            long v = strtol(q, &end, 10);
            if (end == q || v < 0 || v >= count) return BLSMI_E_ARG;
            devs[nalias++] = (int)v;
            q = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') return BLSMI_E_ARG;
        }
    }
    if (nalias) {
        if (ndev <= 0) ndev = nalias;
        if (ndev > nalias) return BLSMI_E_ARG;
        if (g_ready) return (ndev == g_ndev && g_alias) ? BLSMI_OK : BLSMI_E_ARG;
        g_alias = true;
        int rc = ensure_init_list(devs, ndev);
        if (rc) g_alias = false;
        return rc;
    }
    if (ndev <= 0) ndev = count;
    if (ndev > count || ndev > MAX_DEV) return BLSMI_E_ARG;
    if (g_ready) return ndev == g_ndev ? BLSMI_OK : BLSMI_E_ARG;
    for (int i = 0; i < ndev; i++) devs[i] = i;
    return ensure_init_list(devs, ndev);
}
// TEST HOOK, meaningful only under BLSMI_DEVICE_ALIAS: the device-pointer range [p, p + bytes) belongs to logical device `device_index`
// (on real hardware hipPointerGetAttributes answers this; aliased devices share one ordinal).  bytes == 0 forgets the range at p.
BLSMI_API int blsmi_debug_alias_own(const void* p, size_t bytes, int device_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_alias || !p || device_index < 0 || device_index >= g_ndev) return BLSMI_E_ARG;
    for (size_t i = 0; i < g_alias_own.size(); i++) if (g_alias_own[i].lo == (const char*)p) { g_alias_own.erase(g_alias_own.begin() + i); break; }
    if (bytes) g_alias_own.push_back({(const char*)p, (const char*)p + bytes, device_index});
    return BLSMI_OK;
}
// context leases (= entry-point calls and shards of split calls) device `device_index` has served since blsmi_init*; -1: no such device
BLSMI_API long long blsmi_debug_device_leases(int device_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ready || device_index < 0 || device_index >= g_ndev) return -1;
    return (long long)g_dev[device_index].leases;
}
BLSMI_API int blsmi_device_count(void) { std::lock_guard<std::mutex> lk(g_mu); return g_ready ? g_ndev : 0; }
BLSMI_API int blsmi_shard_count(void) { std::lock_guard<std::mutex> lk(g_mu); return g_ready ? g_nshards : 0; }
BLSMI_API void blsmi_shutdown(void) {
    std::unique_lock<std::mutex> lk(g_mu);
    if (!g_ready) return;
    for (;;) {                          // wait for calls in flight
        bool busy = false;
        for (int d = 0; d < g_ndev; d++) for (int i = 0; i < MAX_CTX; i++) busy |= g_dev[d].ctx[i].busy;
        if (!busy) break;
        g_cv.wait(lk);
    }
    for (int d = 0; d < g_ndev; d++) {
        Device& dv = g_dev[d];
        (void)hipSetDevice(dv.id);
        if (dv.comm) { (void)hipStreamSynchronize(dv.coll_stream); (void)g_rccl.CommDestroy(dv.comm); dv.comm = nullptr; }
        if (dv.coll_stream) { (void)hipStreamDestroy(dv.coll_stream); dv.coll_stream = nullptr; }
        dv.coll.release();
        for (int i = 0; i < MAX_CTX; i++) {
            Ctx& c = dv.ctx[i];
            if (!c.stream) continue;
            (void)hipStreamSynchronize(c.stream);
            c.ws.release();
            c.arena.release();
            for (auto& e : c.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            for (auto& e : c.pev) (void)hipEventDestroy(e);
            c.pev.clear(); c.pname.clear(); c.pused = 0;
            for (auto& a : c.aux) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); a = nullptr; }
            if (c.fork) { (void)hipEventDestroy(c.fork); c.fork = nullptr; }
            for (auto& j : c.join) if (j) { (void)hipEventDestroy(j); j = nullptr; }
            (void)hipStreamDestroy(c.stream);
            c.stream = nullptr;
        }
        if (dv.gens.g1) { (void)hipFree(dv.gens.g1); (void)hipFree(dv.gens.g2); (void)hipFree(dv.gens.lines); (void)hipFree(dv.gens.lines_pair); (void)hipFree(dv.gens.lat); (void)hipFree(dv.gens.fixed1); (void)hipFree(dv.gens.fixed2); dv.gens = Gens{}; }
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, dv.id) == hipSuccess) (void)hipMemPoolTrimTo(pool, 0);
    }
    g_have_comm = false;
    g_alias = false; g_alias_own.clear();
    g_ndev = 0;
    g_ready = false;
}
// Give device memory the library holds for FUTURE calls back to the driver: every idle call context's temporaries beyond
// keep_bytes_per_context (0: all of them) and the stream-ordered pool's cache.  Contexts serving a call are skipped.  Tables the caller
// created (blsmi_g2_prepared_create) and the per-device constants are not touched.
BLSMI_API int blsmi_trim(size_t keep_bytes_per_context, size_t* freed_bytes) {
    std::unique_lock<std::mutex> coll_lk(g_coll_mu, std::try_to_lock);
    std::unique_lock<std::mutex> lk(g_mu);
    size_t freed = 0;
    if (freed_bytes) *freed_bytes = 0;
    if (!g_ready) return BLSMI_OK;
    for (int d = 0; d < g_ndev; d++) {
        Device& dv = g_dev[d];
        HIPCHK(hipSetDevice(dv.id));
        for (int i = 0; i < MAX_CTX; i++) {
            Ctx& c = dv.ctx[i];
            if (c.busy) continue;                                          // under g_mu: nobody can lease it meanwhile
            if (c.ws.cap > keep_bytes_per_context) { freed += c.ws.cap; c.ws.release(); }
            freed += c.arena.trim(keep_bytes_per_context > c.ws.cap ? keep_bytes_per_context - c.ws.cap : 0);
        }
        // the exchange buffers belong to whichever split call holds g_coll_mu -- for its whole duration, also in the windows where it holds
        // no context lease (after coll_reserve_all, during the collectives and the copy of the bitmap: ADVICE r04).  try_lock: a trim that
        // meets a split call in flight leaves the exchange buffers alone.
        if (coll_lk.owns_lock() && dv.coll.cap > keep_bytes_per_context) { freed += dv.coll.cap; dv.coll.release(); }
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, dv.id) == hipSuccess) (void)hipMemPoolTrimTo(pool, 0);
    }
    if (tl_ctx) (void)hipSetDevice(tl_ctx->dev->id);
    if (freed_bytes) *freed_bytes = freed;
    return BLSMI_OK;
}
// bytes of device memory the call contexts currently hold for temporaries (all devices); diagnostic companion of blsmi_trim
BLSMI_API size_t blsmi_held_bytes(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    size_t t = 0;
    if (!g_ready) return 0;
    for (int d = 0; d < g_ndev; d++) { for (int i = 0; i < MAX_CTX; i++) t += g_dev[d].ctx[i].arena.total() + g_dev[d].ctx[i].ws.cap; t += g_dev[d].coll.cap; }
    return t;
}
BLSMI_API const char* blsmi_version(void) { return g_version; }

// Page-locked host memory for the host entry points' buffers: hipMemcpyAsync from / to pageable memory is staged by the runtime
// (~10 GB/s), from page-locked memory it is one DMA at PCIe rate.  Portable: every device of the process sees the same mapping.
BLSMI_API int blsmi_host_alloc(size_t bytes, void** out) {
    if (!out) return BLSMI_E_ARG;
    *out = nullptr;
    if (bytes == 0) return BLSMI_OK;
    LOCK_AND_INIT();
    if (hipHostMalloc(out, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return BLSMI_E_NOMEM; }
    return BLSMI_OK;
}
BLSMI_API int blsmi_host_free(void* p) {
    if (!p) return BLSMI_OK;
    if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return BLSMI_E_ARG; }
    return BLSMI_OK;
}

// ---- point formats at the host boundary (blsmi 0.6: the *_jac entry points) --------------------------------------------------
// A Go caller holds its keys and signatures as *bls.G2Projective / *bls.G1Projective (g2pubs/bls.go:13-15, 53-55): x, y, z, each FQ
// 6 LE u64 Montgomery(2^384) limbs (fq.go:11-13, fqrepr.go:14) -- 144 bytes (G1) / 288 bytes (G2), contiguous.  The *_jac entry points
// take those bytes as they lie (one memcpy per point in the shim) and run ToAffine + SerializeBytes on the device (k_wire.hip:
// k_g?_jac_to_affine), after which every path is the affine one.  fmt bit 0: the public keys, bit 1: the signatures arrive that way.
namespace {
constexpr int FMT_PK_JAC = 1, FMT_SIG_JAC = 2, FMT_JAC = 3;
inline size_t rec_bytes(size_t wire_bytes, bool jac) { return jac ? wire_bytes / 2 * 3 : wire_bytes; }
// n in-memory records on the device -> n wire records (the all-zero record for z == 0) and, if asked, their infinity flags
int jac_to_wire_dev(size_t wire_bytes, const void* d_jac, void* d_out, void* d_inf, size_t n, hipStream_t s) {
    if (n == 0) return BLSMI_OK;
    const bool mark = tl_ctx && s == tl_ctx->stream;                       // (profile marks are events on the context's main stream)
    if (mark) prof_mark(wire_bytes == 96 ? "k_g1_jac_to_affine" : "k_g2_jac_to_affine");
    if (wire_bytes == 96) hipLaunchKernelGGL(k_g1_jac_to_affine, dim3(nblocks(n)), dim3(WG), 0, s, (const u64*)d_jac, (u8*)d_out, (u8*)d_inf, n);
    else hipLaunchKernelGGL(k_g2_jac_to_affine, dim3(nblocks(n)), dim3(WG), 0, s, (const u64*)d_jac, (u8*)d_out, (u8*)d_inf, n);
    if (mark) prof_mark(nullptr);
    HIPCHK(hipGetLastError());
    return BLSMI_OK;
}
// host copy + conversion of n records of one group: `host` holds wire records (jac == false: plain copy into d_wire) or in-memory
// records (copied into a temporary of the call, converted into d_wire); everything on s
int upload_points(size_t wire_bytes, bool jac, const uint8_t* host, void* d_wire, size_t n, hipStream_t s) {
    if (!jac) { HIPCHK(hipMemcpyAsync(d_wire, host, wire_bytes * n, hipMemcpyHostToDevice, s)); return BLSMI_OK; }
    DBuf raw; HIPCHK(raw.alloc(rec_bytes(wire_bytes, true) * n));
    HIPCHK(hipMemcpyAsync(raw.p, host, rec_bytes(wire_bytes, true) * n, hipMemcpyHostToDevice, s));
    return jac_to_wire_dev(wire_bytes, raw.p, d_wire, nullptr, n, s);
}
// Is ONE host-side in-memory record the point at infinity?  z.IsZero() (g1.go:287-289, g2.go:325-327), where a coordinate that is not
// below q counts as 0 exactly as on the device (device_io.cuh: load_m384_checked).  No field arithmetic: a compare.
bool jac_host_is_infinity(const uint8_t* rec, size_t wire_bytes) {
    static const uint64_t Q[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    const size_t nc = wire_bytes / 96;                                     // FQ per coordinate
    uint64_t nz = 0;
    for (size_t e = 0; e < nc; e++) {
        uint64_t w[6];
        memcpy(w, rec + 48 * (2 * nc + e), 48);
        bool below = false;
        for (int j = 5; j >= 0; j--) { if (w[j] != Q[j]) { below = w[j] < Q[j]; break; } }
        if (below) for (int j = 0; j < 6; j++) nz |= w[j];
    }
    return nz == 0;
}
}  // namespace

// ---- pairing ------------------------------------------------------------------------------------
static int pairing_dev(const void* d_g1, const void* d_g2, void* d_out, size_t n, hipStream_t s, int mode) {
    if (n == 0) return BLSMI_OK;
    if (mode == 0 && use_lat(n)) {                                         // small call: one pairing per wave (k_lat.hip)
        const size_t prog = g_lat_rolled.load(std::memory_order_relaxed) ? (size_t)LAT_PAIRING1_OFFSET : (size_t)LAT_PAIRING1S_OFFSET;   // the same pairing, its squaring runs unrolled (A/B partner)
        prof_mark(prog == LAT_PAIRING1_OFFSET ? "k_lat:pairing1" : "k_lat:pairing1s");
        hipLaunchKernelGGL(k_lat, dim3((unsigned)n), dim3(64), lat_lds_bytes(prog), s, (const u8*)g_gens.lat + prog,
                           (const u8*)d_g1, (size_t)96, (const u8*)d_g2, (size_t)192, (const u8*)nullptr, (size_t)0, (const u8*)nullptr, (size_t)0,
                           (const u8*)nullptr, (u8*)nullptr, (u64*)d_out, n);
        prof_mark(nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        return BLSMI_OK;
    }
    if (mode == 1 && n <= g_lat_max) {                                     // small MillerLoop call: the reference's steps as a level program
        prof_mark("k_lat:miller1x");
        hipLaunchKernelGGL(k_lat, dim3((unsigned)n), dim3(64), lat_lds_bytes(LAT_MILLER1X_OFFSET), s, (const u8*)g_gens.lat + LAT_MILLER1X_OFFSET,
                           (const u8*)d_g1, (size_t)96, (const u8*)d_g2, (size_t)192, (const u8*)nullptr, (size_t)0, (const u8*)nullptr, (size_t)0,
                           (const u8*)nullptr, (u8*)nullptr, (u64*)d_out, n);
        prof_mark(nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        return BLSMI_OK;
    }
    HIPCHK(g_ws.reserve(sizeof(i32) * 12 * NL * n));
    i32* f = reinterpret_cast<i32*>(g_ws.p);
    const unsigned pblocks = (unsigned)((n + PT - 1) / PT);
    if (mode == 0 && use_row(n, true)) {                                   // a few thousand tuples: sixteen lanes per tuple, one wave per SIMD at 4 096 tuples
        prof_mark("k_miller1h_row");
        hipLaunchKernelGGL(k_miller1h_row, dim3(rblocks(n)), dim3(WG), 0, s, (const u8*)d_g1, (const u8*)d_g2, f, n);
        prof_mark("k_final_exp_row");
        hipLaunchKernelGGL(k_final_exp_row, dim3(rblocks(n)), dim3(WG), 0, s, (const i32*)f, (u64*)d_out, n, 0);
        prof_mark(nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        return BLSMI_OK;
    }
    if (mode == 0 && use_quad(n)) {                                        // mid-size batch: four lanes per tuple, one wave per SIMD at 16 384 tuples
        prof_mark("k_miller1h_quad");
        hipLaunchKernelGGL(k_miller1h_quad, dim3(qblocks(n)), dim3(WG), 0, s, (const u8*)d_g1, (const u8*)d_g2, f, n);
        prof_mark("k_final_exp_quad");
        hipLaunchKernelGGL(k_final_exp_quad, dim3(qblocks(n)), dim3(WG), 0, s, (const i32*)f, (u64*)d_out, n, 0);
        prof_mark(nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        return BLSMI_OK;
    }
    // mode 1 (MillerLoop): the reference's steps; mode 0 (Pairing): the homogeneous steps
    prof_mark(g_pair_layout ? (mode ? "k_miller1_pair" : "k_miller1h_pair") : (mode ? "k_miller1" : "k_miller1h"));
    if (g_pair_layout) hipLaunchKernelGGL(mode ? k_miller1_pair : k_miller1h_pair, dim3(pblocks), dim3(WG), 0, s, (const u8*)d_g1, (const u8*)d_g2, f, n);
    else hipLaunchKernelGGL(mode ? k_miller1 : k_miller1h, dim3(nblocks(n)), dim3(WG), 0, s, (const u8*)d_g1, (const u8*)d_g2, f, n);
    prof_mark(g_pair_layout ? "k_final_exp_pair" : "k_final_exp");
    if (g_pair_layout) hipLaunchKernelGGL(k_final_exp_pair, dim3(pblocks), dim3(WG), 0, s, (const i32*)f, (u64*)d_out, n, mode);
    else hipLaunchKernelGGL(k_final_exp, dim3(nblocks(n)), dim3(WG), 0, s, (const i32*)f, (u64*)d_out, n, mode);
    prof_mark(nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));      // blocking entry point: results are ready on return
    return BLSMI_OK;
}
// Enable/disable per-kernel HIP-event timing of blsmi_pairing_batch[_dev]; read back (by the calling thread, for its own
// last call) with blsmi_last_kernel_ms.
// Batches of at most `max_tuples` tuples take the latency path (one tuple per wave, k_lat.hip); 0 switches it off.
BLSMI_API int blsmi_set_latency_threshold(size_t max_tuples) {
    set_explicit(X_LAT_MAX);
    g_lat_max.store(max_tuples);
    return BLSMI_OK;
}
// Latency hint for the host shim: 1 when n operations of `shape` in ONE call are expected to finish sooner on one core of the upstream
// pure-Go path than on the device.  A lone call costs the depth of one wave walking the whole computation (flat in n up to ~2 000
// operations), so the break-even is n* = device latency of a lone call / CPU time per operation -- both measured by bench.py's
// `reference_shapes` leg (profiles/r04*_bench_detail.json: one MI355X against one host core on the C restatement of the reference).
BLSMI_API int blsmi_prefer_cpu(int shape, size_t n) {
    struct Row { float gpu_ms_lone_call, cpu_ms_per_op; };
    static const Row rows[] = {
        /* BLSMI_SHAPE_PAIRING         */ {1.38f, 2.92f},
        /* BLSMI_SHAPE_MILLER_LOOP     */ {0.71f, 0.54f},      // over an already prepared G2 argument, as bls.MillerLoop takes it
        /* BLSMI_SHAPE_FINAL_EXP       */ {0.84f, 2.14f},
        /* BLSMI_SHAPE_G2_PREPARE      */ {1.39f, 0.19f},
        /* BLSMI_SHAPE_VERIFY          */ {2.19f, 3.72f},
        /* BLSMI_SHAPE_SIGN            */ {1.43f, 0.45f},
        /* BLSMI_SHAPE_VERIFY_DOMAIN   */ {3.18f, 5.62f},
        /* BLSMI_SHAPE_POINT_ADD       */ {0.25f, 0.0065f},    // AggregateSignatures / AggregatePublicKeys: one Jacobian addition per element
    };
    if (shape < 0 || shape >= (int)(sizeof rows / sizeof rows[0])) return 0;
    if (n == 0) return 1;
    return (double)n * rows[shape].cpu_ms_per_op < rows[shape].gpu_ms_lone_call ? 1 : 0;
}
BLSMI_API int blsmi_set_quad_threshold(size_t max_tuples) {
    set_explicit(X_QUAD_MAX);
    g_quad_max.store(max_tuples);
    return BLSMI_OK;
}
BLSMI_API int blsmi_set_row_threshold(size_t min_tuples, size_t max_tuples) {
    set_explicit(X_ROW);
    g_row_min.store(min_tuples);
    g_row_max.store(max_tuples);
    return BLSMI_OK;
}
BLSMI_API int blsmi_set_mul_assume_subgroup(int on) {
    set_explicit(X_MUL_SUBGROUP);
    g_mul_subgroup.store(on != 0);
    return BLSMI_OK;
}
// Run-time switches of code paths that produce identical results (A/B tests, soaks): name = "agg_cofactor_pow" (large g2pubs aggregates
// raise their Miller product to 1 - x instead of clearing n hash points), "msm_sort" (device radix sort against the exact histogram passes),
// "dup_force_sort" (the duplicate screen's host-sort fallback on every call).  value: 0 / 1.  Atomic: a call in flight sees the old or the new
// value, never a torn one; the verdicts do not depend on them.
BLSMI_API int blsmi_set_option(const char* name, long long value) {
    if (!name) return BLSMI_E_ARG;
    const std::string n(name);
    if (n == "agg_cofactor_pow") { set_explicit(X_AGG_POW); g_agg_cofactor_pow.store(value != 0); }
    else if (n == "msm_sort") { set_explicit(X_MSM_SORT); g_msm_sort.store(value != 0); }
    else if (n == "dup_force_sort") { set_explicit(X_DUP_SORT); g_dup_force_sort.store(value != 0); }
    else if (n == "lat_rolled") { set_explicit(X_LAT_ROLLED); g_lat_rolled.store(value != 0); }
    else if (n == "crowd_quad") { set_explicit(X_CROWD_QUAD); g_crowd_quad.store(value != 0); }
    else if (n == "row_side") { set_explicit(X_ROW_SIDE); g_row_side.store(value != 0); }
    else if (n == "row_side_g2pubs") g_row_side_g2pubs.store(value != 0);
    else if (n == "row_side_lds") g_row_side_lds.store((size_t)std::min(65536LL, std::max(0LL, value)));
    else if (n == "row_side_piece") g_row_side_piece.store((size_t)std::max(0LL, value));
    else if (n == "hash_row_min") g_hash_row_min.store((size_t)std::max(0LL, value));
    else if (n == "hash_row_max") g_hash_row_max.store((size_t)std::max(0LL, value));
    else if (n == "swu_row_max") g_swu_row_max.store((size_t)std::max(0LL, value));
    else if (n == "hash_oct_min") g_hash_oct_min.store((size_t)std::max(0LL, value));
    else if (n == "hash_oct_max") g_hash_oct_max.store((size_t)std::max(0LL, value));
    else if (n == "hash_g1_quad_min") g_hash_g1_quad_min.store((size_t)std::max(0LL, value));
    else if (n == "hash_g1_quad_max") g_hash_g1_quad_max.store((size_t)std::max(0LL, value));
    else if (n == "hash_quad_min") g_hash_quad_min.store((size_t)std::max(0LL, value));
    else if (n == "hash_quad_max") g_hash_quad_max.store((size_t)std::max(0LL, value));
    else if (n == "combine_mid_max") { set_explicit(X_COMBINE_MID); g_combine_mid_max.store((size_t)std::max(0LL, value)); }
    else if (n == "crowd_floor") { set_explicit(X_CROWD_FLOOR); g_crowd_floor.store((size_t)std::max(0LL, value)); }
    else if (n == "assume_load") g_assume_load.store((size_t)std::max(0LL, value));
    else return BLSMI_E_ARG;
    return BLSMI_OK;
}
BLSMI_API int blsmi_set_profiling(int on) {
    g_profile.store(on != 0);
    return BLSMI_OK;
}
// the calling thread's profile log as "name=ms;name=ms;..." (kernels in launch order; a name may repeat), read-and-clear.
// Returns the length needed (excluding the terminator); the text is truncated to cap - 1 characters.
BLSMI_API int blsmi_last_profile(char* out, size_t cap) {
    std::string t;
    char tmp[96];
    for (auto& p : tl_prof) { snprintf(tmp, sizeof tmp, "%s=%.4f;", p.first, p.second); t += tmp; }
    tl_prof.clear();
    if (out && cap) { const size_t k = std::min(cap - 1, t.size()); memcpy(out, t.data(), k); out[k] = 0; }
    return (int)t.size();
}
// Miller-loop / final-exponentiation kernel times of the calling thread's last profiled blsmi_pairing_batch[_dev] call
// (kept for callers of blsmi 0.2; reads the same log without clearing it)
BLSMI_API int blsmi_last_kernel_ms(float* miller_ms, float* final_exp_ms) {
    if (!miller_ms || !final_exp_ms) return BLSMI_E_ARG;
    for (auto& p : tl_prof) {
        if (!strncmp(p.first, "k_miller1", 9)) tl_last_ms[0] = p.second;
        else if (!strncmp(p.first, "k_final_exp", 11)) tl_last_ms[1] = p.second;
    }
    *miller_ms = tl_last_ms[0]; *final_exp_ms = tl_last_ms[1];
    return BLSMI_OK;
}
BLSMI_API int blsmi_pairing_batch_dev(const void* d_g1, const void* d_g2, void* d_out, size_t n, void* stream) {
    if (n == 0) return BLSMI_OK;
    if (!d_g1 || !d_g2 || !d_out) return BLSMI_E_ARG;
    LOCK_AND_INIT_AT(d_out);
    UseStream us(stream);
    return pairing_dev(d_g1, d_g2, d_out, n, g_stream, 0);
}
// host-buffer form: independent tuples, split over the devices by contiguous block, no exchange at all
static int pairing_host(const uint8_t* g1, const uint8_t* g2, uint64_t* out, size_t n, int mode, bool jac = false) {
    if (n && (!g1 || !g2 || !out)) return BLSMI_E_ARG;
    if (n == 0) return BLSMI_OK;
    { std::lock_guard<std::mutex> lk(g_mu); int rc = ensure_init_default(); if (rc) return rc; }
    return run_shards(plan_shards(n, 64), [&](int, size_t lo, size_t hi) -> int {
        const size_t m = hi - lo;
        DBuf a, b, o;
        HIPCHK(a.alloc(96 * m)); HIPCHK(b.alloc(192 * m)); HIPCHK(o.alloc(576 * m));
        int rc = upload_points(96, jac, g1 + rec_bytes(96, jac) * lo, a.p, m, g_stream);
        if (rc) return rc;
        rc = upload_points(192, jac, g2 + rec_bytes(192, jac) * lo, b.p, m, g_stream);
        if (rc) return rc;
        rc = pairing_dev(a.p, b.p, o.p, m, g_stream, mode);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(out + 72 * lo, o.p, 576 * m, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        return BLSMI_OK;
    });
}
BLSMI_API int blsmi_pairing_batch(const uint8_t* g1, const uint8_t* g2, uint64_t* out, size_t n) { return pairing_host(g1, g2, out, n, 0); }
BLSMI_API int blsmi_miller_loop_batch(const uint8_t* g1, const uint8_t* g2, uint64_t* out, size_t n) { return pairing_host(g1, g2, out, n, 1); }
// bls.Pairing(p *G1Projective, q *G2Projective) (pairing.go:132-136) on the points as the Go heap holds them: ToAffine on the device
BLSMI_API int blsmi_pairing_batch_jac(const uint64_t* g1_jac, const uint64_t* g2_jac, uint64_t* out, size_t n) {
    return pairing_host(reinterpret_cast<const uint8_t*>(g1_jac), reinterpret_cast<const uint8_t*>(g2_jac), out, n, 0, true);
}
// ... and with the in-memory points RESIDENT on one of the library's devices (a verifier that keeps its keys in HBM as it holds them in Go)
BLSMI_API int blsmi_pairing_batch_jac_dev(const void* d_g1_jac, const void* d_g2_jac, void* d_out, size_t n, void* stream) {
    if (n == 0) return BLSMI_OK;
    if (!d_g1_jac || !d_g2_jac || !d_out) return BLSMI_E_ARG;
    LOCK_AND_INIT_AT(d_out);
    UseStream us(stream);
    DBuf a, b;
    HIPCHK(a.alloc(96 * n)); HIPCHK(b.alloc(192 * n));
    int rc = jac_to_wire_dev(96, d_g1_jac, a.p, nullptr, n, g_stream);
    if (!rc) rc = jac_to_wire_dev(192, d_g2_jac, b.p, nullptr, n, g_stream);
    if (rc) return rc;
    return pairing_dev(a.p, b.p, d_out, n, g_stream, 0);
}
// G?Projective.ToAffine().SerializeBytes() (g1.go:322-340 + 157-167, g2.go:365-386 + 172-186) for n points: wire records (all zero for
// the point at infinity) and the infinity flags
template <int PB>
static int jac_to_affine_host(const uint64_t* jac, uint8_t* out, uint8_t* out_inf, size_t n) {
    if (n && (!jac || !out)) return BLSMI_E_ARG;
    if (n == 0) return BLSMI_OK;
    { std::lock_guard<std::mutex> lk(g_mu); int rc = ensure_init_default(); if (rc) return rc; }
    return run_shards(plan_shards(n, 64), [&](int, size_t lo, size_t hi) -> int {
        const size_t m = hi - lo, in = rec_bytes(PB, true);
        DBuf raw, o, fl;
        HIPCHK(raw.alloc(in * m)); HIPCHK(o.alloc((size_t)PB * m)); HIPCHK(fl.alloc(m));
        HIPCHK(hipMemcpyAsync(raw.p, reinterpret_cast<const uint8_t*>(jac) + in * lo, in * m, hipMemcpyHostToDevice, g_stream));
        int rc = jac_to_wire_dev(PB, raw.p, o.p, fl.p, m, g_stream);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(out + (size_t)PB * lo, o.p, (size_t)PB * m, hipMemcpyDeviceToHost, g_stream));
        if (out_inf) HIPCHK(hipMemcpyAsync(out_inf + lo, fl.p, m, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        return BLSMI_OK;
    });
}
BLSMI_API int blsmi_g1_jac_to_affine_batch(const uint64_t* jac, uint8_t* out, uint8_t* out_inf, size_t n) { return jac_to_affine_host<96>(jac, out, out_inf, n); }
BLSMI_API int blsmi_g2_jac_to_affine_batch(const uint64_t* jac, uint8_t* out, uint8_t* out_inf, size_t n) { return jac_to_affine_host<192>(jac, out, out_inf, n); }
BLSMI_API int blsmi_final_exponentiation_batch(const uint64_t* in, uint64_t* out, size_t n) {
    if (n && (!in || !out)) return BLSMI_E_ARG;
    LOCK_AND_INIT();
    if (n == 0) return BLSMI_OK;
    DBuf i, f, o;
    HIPCHK(i.alloc(576 * n)); HIPCHK(o.alloc(576 * n)); HIPCHK(f.alloc(sizeof(i32) * 12 * NL * n));
    HIPCHK(hipMemcpyAsync(i.p, in, 576 * n, hipMemcpyHostToDevice, g_stream));
    if (use_row(n)) {                                                      // a few thousand values: sixteen lanes per value (k_pairing_row.hip), 1.2 ms up to 4 096
        hipLaunchKernelGGL(k_fq12_from_m384, dim3(nblocks(n)), dim3(WG), 0, g_stream, i.as<u64>(), f.as<i32>(), n);
        hipLaunchKernelGGL(k_final_exp_row, dim3(rblocks(n)), dim3(WG), 0, g_stream, (const i32*)f.as<i32>(), o.as<u64>(), n, 0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, o.p, 576 * n, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        return BLSMI_OK;
    }
    if (n <= g_lat_max) {                                                  // small call: one final exponentiation per wave (k_lat.hip)
        hipLaunchKernelGGL(k_lat, dim3((unsigned)n), dim3(64), lat_lds_bytes(LAT_FINALEXP1_OFFSET), g_stream, (const u8*)g_gens.lat + LAT_FINALEXP1_OFFSET,
                           (const u8*)i.p, (size_t)576, (const u8*)nullptr, (size_t)0, (const u8*)nullptr, (size_t)0, (const u8*)nullptr, (size_t)0,
                           (const u8*)nullptr, (u8*)nullptr, o.as<u64>(), n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, o.p, 576 * n, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        return BLSMI_OK;
    }
    hipLaunchKernelGGL(k_fq12_from_m384, dim3(nblocks(n)), dim3(WG), 0, g_stream, i.as<u64>(), f.as<i32>(), n);
    hipLaunchKernelGGL(k_final_exp, dim3(nblocks(n)), dim3(WG), 0, g_stream, f.as<i32>(), o.as<u64>(), n, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, o.p, 576 * n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return BLSMI_OK;
}

// ---- unit-level ops -------------------------------------------------------------------------------
BLSMI_API int blsmi_debug_op(int op_in, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flag, size_t n) {
    const bool pairl = (op_in & BLSMI_OP_LANE_PAIR) != 0;                 // run the tower op in the lane-pair layout
    const bool quadl = (op_in & BLSMI_OP_LANE_QUAD) != 0;                 // ... the Fq12 op in the lane-quad layout
    const bool rowl = (op_in & BLSMI_OP_LANE_ROW) != 0;                   // ... in the lane-row layout
    const int op = op_in & ~(BLSMI_OP_LANE_PAIR | BLSMI_OP_LANE_QUAD | BLSMI_OP_LANE_ROW);
    if (pairl && (op < 16 || op >= 64)) return BLSMI_E_ARG;
    const bool quadg2 = quadl && op >= BLSMI_OP_ROW_G2_DOUBLE && op <= BLSMI_OP_ROW_CLEAR_H2;   // quad_g2.inc
    if (quadl && !quadg2 && (pairl || rowl || op < BLSMI_OP_FQ12_MUL || op > BLSMI_OP_FQ12_MUL_BY_014)) return BLSMI_E_ARG;
    if (quadg2 && (pairl || rowl)) return BLSMI_E_ARG;
    const bool rowstep = op >= BLSMI_OP_ROW_DBL_STEP && op <= BLSMI_OP_ROW_CLEAR_H2;
    if (rowstep && !rowl && !quadg2) return BLSMI_E_ARG;
    if (rowl && !rowstep && (pairl || op < BLSMI_OP_FQ12_MUL || op > BLSMI_OP_FQ12_MUL_BY_014 || op == BLSMI_OP_FQ12_CYCLO_RUN16)) return BLSMI_E_ARG;
    int width = op < 16 ? 1 : op < 32 ? 2 : op < 48 ? 6 : op < 64 ? 12 : rowstep ? 12 : (op == BLSMI_OP_G1_DOUBLE || op == BLSMI_OP_G1_ADD || op == BLSMI_OP_SWU_G1) ? 3 : 6;
    if (n && (!a || !out)) return BLSMI_E_ARG;
    LOCK_AND_INIT();
    if (n == 0) return BLSMI_OK;
    const size_t bytes = (size_t)48 * width * n;
    DBuf da, db, dout, dflag;
    HIPCHK(da.alloc(bytes)); HIPCHK(db.alloc(bytes)); HIPCHK(dout.alloc(bytes)); HIPCHK(dflag.alloc(n));
    HIPCHK(hipMemcpyAsync(da.p, a, bytes, hipMemcpyHostToDevice, g_stream));
    if (b) HIPCHK(hipMemcpyAsync(db.p, b, bytes, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemsetAsync(dflag.p, 1, n, g_stream));
    dim3 g(nblocks(n)), w(WG);
    if (rowl) hipLaunchKernelGGL(k_debug_row, dim3(rblocks(n)), w, 0, g_stream, op, da.as<u64>(), b ? db.as<u64>() : (const u64*)nullptr, dout.as<u64>(), n);
    else if (quadg2) hipLaunchKernelGGL(k_debug_quad_g2, dim3(qblocks(n)), w, 0, g_stream, op, da.as<u64>(), dout.as<u64>(), n);
    else if (quadl) hipLaunchKernelGGL(k_debug_quad, dim3(qblocks(n)), w, 0, g_stream, op, da.as<u64>(), b ? db.as<u64>() : (const u64*)nullptr, dout.as<u64>(), n);
    else if (pairl) hipLaunchKernelGGL(k_debug_pairl, dim3((unsigned)((n + WG / 2 - 1) / (WG / 2))), w, 0, g_stream, op, da.as<u64>(), b ? db.as<u64>() : (const u64*)nullptr, dout.as<u64>(), n);
    else if (op < 16) hipLaunchKernelGGL(k_debug_fq, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), dflag.as<u8>(), n);
    else if (op < 32) hipLaunchKernelGGL(k_debug_fq2, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), dflag.as<u8>(), n);
    else if (op < 48) hipLaunchKernelGGL(k_debug_fq6, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), n);
    else if (op < 64) hipLaunchKernelGGL(k_debug_fq12, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), n);
    else if (op == BLSMI_OP_SWU_G1) hipLaunchKernelGGL(k_debug_swu_g1, g, w, 0, g_stream, da.as<u64>(), dout.as<u64>(), n);
    else if (op == BLSMI_OP_SWU_G2) hipLaunchKernelGGL(k_debug_swu_g2, g, w, 0, g_stream, da.as<u64>(), dout.as<u64>(), n);
    else hipLaunchKernelGGL(k_debug_curve, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, g_stream));
    if (flag) HIPCHK(hipMemcpyAsync(flag, dflag.p, n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return BLSMI_OK;
}

// G2Prepared of one point as 68 x 3 Fq2 in the Fq wire format (compare with g2.go:650-801)
BLSMI_API int blsmi_debug_g2_prepare(const uint8_t* g2_aff, int mode, uint64_t* out) {
    if (!out || (mode != 2 && !g2_aff) || mode < 0 || mode > 2) return BLSMI_E_ARG;
    LOCK_AND_INIT();
    DBuf dq, tab, dout;
    const size_t words = (size_t)68 * 3 * 2 * NL;
    HIPCHK(dq.alloc(192)); HIPCHK(tab.alloc(sizeof(i32) * words)); HIPCHK(dout.alloc(sizeof(u64) * 68 * 3 * 12));
    const i32* src = g_gens.lines;
    if (mode != 2) {
        HIPCHK(hipMemcpyAsync(dq.p, g2_aff, 192, hipMemcpyHostToDevice, g_stream));
        if (mode == 0) hipLaunchKernelGGL(k_debug_prepare_single, dim3(1), dim3(WG), 0, g_stream, dq.as<u8>(), tab.as<i32>());
        else hipLaunchKernelGGL(k_debug_prepare_pair, dim3(1), dim3(WG), 0, g_stream, dq.as<u8>(), tab.as<i32>());
        src = tab.as<i32>();
    }
    hipLaunchKernelGGL(k_debug_lines_to_m384, dim3((68 * 3 * 2 + WG - 1) / WG), dim3(WG), 0, g_stream, src, dout.as<u64>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, sizeof(u64) * 68 * 3 * 12, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return BLSMI_OK;
}

// ---- scalar multiplication / sums ----------------------------------------------------------------
// d_pts == nullptr: every scalar multiplies the group generator (PrivToPub, g2pubs/bls.go:138-140, g1pubs/bls.go:144-146).
// Everything on the device of the leased context; enqueues on s, no synchronisation.
template <int PB, class K>
static int mul_dev_core(K kernel, const u8* d_pts, int gen_group, const u8* d_scalars, u8* d_out, u8* d_inf, size_t m, hipStream_t s) {
    const u8* d_gen = gen_group == 1 ? g_gens.g1 : g_gens.g2;            // the leased device's copy of the generator
    const u8* base = d_pts ? d_pts : d_gen;
    const size_t stride = d_pts ? (size_t)PB : 0;
    if (!d_pts) {
        // the generator as the common multiplicand (PrivToPub): the device's fixed-base table -- 32 mixed additions, no doublings.
        // Small calls: one scalar per wave (the 32 table entries meet in a tree of additions across the lanes).
        const i32* table = PB == 96 ? g_gens.fixed1 : g_gens.fixed2;
        if (m <= g_env.fixed_wave_max && m <= g_lat_max) {
            prof_mark(PB == 96 ? "k_g1_mul_fixed_wave" : "k_g2_mul_fixed_wave");
            if (PB == 96) hipLaunchKernelGGL(k_g1_mul_fixed_wave, dim3((unsigned)m), dim3(WG), 0, s, table, d_scalars, d_out, d_inf, m);
            else hipLaunchKernelGGL(k_g2_mul_fixed_wave, dim3((unsigned)m), dim3(WG), 0, s, table, d_scalars, d_out, d_inf, m);
        } else {
            prof_mark(PB == 96 ? "k_g1_mul_fixed" : "k_g2_mul_fixed");
            if (PB == 96) hipLaunchKernelGGL(k_g1_mul_fixed, dim3(nblocks(m)), dim3(WG), 0, s, table, d_scalars, d_out, d_inf, m);
            else hipLaunchKernelGGL(k_g2_mul_fixed, dim3(nblocks(m)), dim3(WG), 0, s, table, d_scalars, d_out, d_inf, m);
        }
        prof_mark(nullptr);
        HIPCHK(hipGetLastError());
        return BLSMI_OK;
    }
    if (m <= g_lat_max && mul_subgroup()) { // small call: one multiplication per wave (k_lat.hip, SEL levels over the decomposed scalar)
        const size_t prog = PB == 96 ? LAT_MUL1_OFFSET : LAT_MUL2_OFFSET;
        DBuf good, rec; HIPCHK(good.alloc(m, s)); HIPCHK(rec.alloc(64 * m, s));
        hipLaunchKernelGGL(k_glv_recode, dim3(nblocks(m)), dim3(WG), 0, s, d_scalars, PB == 96 ? 1 : 2, rec.as<u8>(), m);
        prof_mark(PB == 96 ? "k_lat:mul1" : "k_lat:mul2");
        hipLaunchKernelGGL(k_lat, dim3((unsigned)m), dim3(64), lat_lds_bytes(prog), s, (const u8*)g_gens.lat + prog, base, stride,
                           (const u8*)rec.as<u8>(), (size_t)64, (const u8*)nullptr, (size_t)0, (const u8*)nullptr, (size_t)0,
                           (const u8*)nullptr, good.as<u8>(), reinterpret_cast<u64*>(d_out), m);
        prof_mark(nullptr);
        hipLaunchKernelGGL(k_mul_finish, dim3(nblocks(m)), dim3(WG), 0, s, (const u8*)good.as<u8>(), base, stride, PB / 4, d_out, d_inf, m);
        HIPCHK(hipGetLastError());
        return BLSMI_OK;                                                   // `good` is released in stream order
    }
    // points of the prime-order subgroup (the default): the ladder through the curve endomorphisms (glv.cuh) -- half / a quarter
    // of the doublings; arbitrary curve points (blsmi_set_mul_assume_subgroup(0)): the plain fixed-window ladder
    const bool glv = mul_subgroup();
    if (glv) {
        prof_mark(PB == 192 ? (g_pair_layout ? "k_g2_mul_glv_pair" : "k_g2_mul_glv") : "k_g1_mul_glv");
        if (PB == 192 && g_pair_layout) hipLaunchKernelGGL(k_g2_mul_glv_pair, dim3((unsigned)((m + PT - 1) / PT)), dim3(WG), 0, s, base, stride, d_scalars, d_out, d_inf, m);
        else if (PB == 192) hipLaunchKernelGGL(k_g2_mul_glv, dim3(nblocks(m)), dim3(WG), 0, s, base, stride, d_scalars, d_out, d_inf, m);
        else hipLaunchKernelGGL(k_g1_mul_glv, dim3(nblocks(m)), dim3(WG), 0, s, base, stride, d_scalars, d_out, d_inf, m);
        prof_mark(nullptr);
        HIPCHK(hipGetLastError());
        return BLSMI_OK;
    }
    prof_mark(PB == 192 ? (g_pair_layout ? "k_g2_mul_pair" : "k_g2_mul") : "k_g1_mul");
    if (PB == 192 && g_pair_layout)                                        // G2: lane-pair kernel, two waves per SIMD
        hipLaunchKernelGGL(k_g2_mul_pair, dim3((unsigned)((m + PT - 1) / PT)), dim3(WG), 0, s, base, stride, d_scalars, d_out, d_inf, m);
    else
        hipLaunchKernelGGL(kernel, dim3(nblocks(m)), dim3(WG), 0, s, base, stride, d_scalars, d_out, d_inf, m);
    prof_mark(nullptr);
    HIPCHK(hipGetLastError());
    return BLSMI_OK;
}
template <int PB, class K>
static int mul_batch(K kernel, const uint8_t* pts, int gen_group, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n, bool any_point = false, uint64_t* out_jac = nullptr) {
    // out_jac (replaces out / out_inf): the results as in-memory Jacobian records with z = 1, (0, 1, 0) for the point at infinity
    if (n && ((!pts && !gen_group) || !scalars || (out_jac ? false : (!out || !out_inf)))) return BLSMI_E_ARG;
    if (n == 0) return BLSMI_OK;
    { std::lock_guard<std::mutex> lk(g_mu); int rc = ensure_init_default(); if (rc) return rc; }
    // independent multiplications: large host-buffer batches are split by contiguous block like a verify batch (devices,
    // logical shards): each block has its own host thread and stream, so one block's copies run beside another's kernel
    return run_shards(plan_shards(n, 64), [&](int, size_t lo, size_t hi) -> int {
        const size_t m = hi - lo;
        MulAny any_guard(any_point);                                       // shards run on their own threads
        DBuf dp, ds, dout, dinf;
        HIPCHK(ds.alloc(32 * m)); HIPCHK(dout.alloc((size_t)PB * m)); HIPCHK(dinf.alloc(m));
        if (pts) { HIPCHK(dp.alloc((size_t)PB * m)); HIPCHK(hipMemcpyAsync(dp.p, pts + (size_t)PB * lo, (size_t)PB * m, hipMemcpyHostToDevice, g_stream)); }
        HIPCHK(hipMemcpyAsync(ds.p, scalars + 32 * lo, 32 * m, hipMemcpyHostToDevice, g_stream));
        int rc = mul_dev_core<PB>(kernel, pts ? dp.as<u8>() : nullptr, gen_group, ds.as<u8>(), dout.as<u8>(), dinf.as<u8>(), m, g_stream);
        if (rc) return rc;
        if (out_jac) {
            const size_t jb = rec_bytes(PB, true);
            DBuf dj; HIPCHK(dj.alloc(jb * m));
            hipLaunchKernelGGL(k_affine_to_jac, dim3(nblocks(m)), dim3(WG), 0, g_stream, (const u8*)dout.as<u8>(), (const void*)dinf.p, 1, PB == 96 ? 1 : 2, dj.as<u64>(), m);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(out_jac) + jb * lo, dj.p, jb * m, hipMemcpyDeviceToHost, g_stream));
            HIPCHK(hipStreamSynchronize(g_stream));
            return BLSMI_OK;
        }
        HIPCHK(hipMemcpyAsync(out + (size_t)PB * lo, dout.p, (size_t)PB * m, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipMemcpyAsync(out_inf + lo, dinf.p, m, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        return BLSMI_OK;
    });
}
BLSMI_API int blsmi_g1_mul_batch(const uint8_t* pts, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) { if (n && !pts) return BLSMI_E_ARG; return mul_batch<96>(k_g1_mul, pts, 0, scalars, out, out_inf, n); }
BLSMI_API int blsmi_g2_mul_batch(const uint8_t* pts, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) { if (n && !pts) return BLSMI_E_ARG; return mul_batch<192>(k_g2_mul, pts, 0, scalars, out, out_inf, n); }
BLSMI_API int blsmi_g1_mul_generator_batch(const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) { return mul_batch<96>(k_g1_mul, nullptr, 1, scalars, out, out_inf, n); }
BLSMI_API int blsmi_g2_mul_generator_batch(const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) { return mul_batch<192>(k_g2_mul, nullptr, 2, scalars, out, out_inf, n); }
// PrivToPub for n keys with the results as the Go types hold them (bls.G?Projective with z = 1): PublicKey{p} is the record, no FQReprToFQ on the host
BLSMI_API int blsmi_g1_mul_generator_batch_jac(const uint8_t* scalars, uint64_t* out_jac, size_t n) { return mul_batch<96>(k_g1_mul, nullptr, 1, scalars, nullptr, nullptr, n, false, out_jac); }
BLSMI_API int blsmi_g2_mul_generator_batch_jac(const uint8_t* scalars, uint64_t* out_jac, size_t n) { return mul_batch<192>(k_g2_mul, nullptr, 2, scalars, nullptr, nullptr, n, false, out_jac); }
// device-pointer forms: points (NULL = the group generator), scalars, results and infinity bytes resident on one device
template <int PB, class K>
static int mul_batch_dev(K kernel, int gen_group, const void* d_pts, const void* d_scalars, void* d_out, void* d_out_inf, size_t n, void* stream, bool any_point = false) {
    if (n == 0) return BLSMI_OK;
    if (!d_scalars || !d_out || !d_out_inf) return BLSMI_E_ARG;
    LOCK_AND_INIT_AT(d_out);
    UseStream us(stream);
    MulAny any_guard(any_point);
    int rc = mul_dev_core<PB>(kernel, (const u8*)d_pts, gen_group, (const u8*)d_scalars, (u8*)d_out, (u8*)d_out_inf, n, g_stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g_stream));
    return BLSMI_OK;
}
BLSMI_API int blsmi_g1_mul_batch_dev(const void* d_pts, const void* d_scalars, void* d_out, void* d_out_inf, size_t n, void* stream) { return mul_batch_dev<96>(k_g1_mul, 1, d_pts, d_scalars, d_out, d_out_inf, n, stream); }
BLSMI_API int blsmi_g2_mul_batch_dev(const void* d_pts, const void* d_scalars, void* d_out, void* d_out_inf, size_t n, void* stream) { return mul_batch_dev<192>(k_g2_mul, 2, d_pts, d_scalars, d_out, d_out_inf, n, stream); }
// per-call choice of the ladder (flags & BLSMI_MUL_ANY_POINT: the plain windowed ladder, which serves every curve point like MulFR, g1.go:80-90)
BLSMI_API int blsmi_g1_mul_batch_ex(const uint8_t* pts, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n, unsigned flags) {
    if ((n && !pts) || (flags & ~(unsigned)BLSMI_MUL_ANY_POINT)) return BLSMI_E_ARG;
    return mul_batch<96>(k_g1_mul, pts, 0, scalars, out, out_inf, n, (flags & BLSMI_MUL_ANY_POINT) != 0);
}
BLSMI_API int blsmi_g2_mul_batch_ex(const uint8_t* pts, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n, unsigned flags) {
    if ((n && !pts) || (flags & ~(unsigned)BLSMI_MUL_ANY_POINT)) return BLSMI_E_ARG;
    return mul_batch<192>(k_g2_mul, pts, 0, scalars, out, out_inf, n, (flags & BLSMI_MUL_ANY_POINT) != 0);
}
BLSMI_API int blsmi_g1_mul_batch_dev_ex(const void* d_pts, const void* d_scalars, void* d_out, void* d_out_inf, size_t n, void* stream, unsigned flags) {
    if ((n && !d_pts) || (flags & ~(unsigned)BLSMI_MUL_ANY_POINT)) return BLSMI_E_ARG;
    return mul_batch_dev<96>(k_g1_mul, 1, d_pts, d_scalars, d_out, d_out_inf, n, stream, (flags & BLSMI_MUL_ANY_POINT) != 0);
}
BLSMI_API int blsmi_g2_mul_batch_dev_ex(const void* d_pts, const void* d_scalars, void* d_out, void* d_out_inf, size_t n, void* stream, unsigned flags) {
    if ((n && !d_pts) || (flags & ~(unsigned)BLSMI_MUL_ANY_POINT)) return BLSMI_E_ARG;
    return mul_batch_dev<192>(k_g2_mul, 2, d_pts, d_scalars, d_out, d_out_inf, n, stream, (flags & BLSMI_MUL_ANY_POINT) != 0);
}

// tree reduction of n affine points already on the device; result (affine bytes + inf flag) on the device
template <int PB, int W, class K0, class K1, class K2>
static int sum_dev(K0 k0, K1 k1, K2 kfinal, const u8* d_pts, const u8* d_inf, size_t n, u8* d_out, i32* d_out_inf, hipStream_t s, bool sync = true, bool jac = false) {
    const size_t words = (size_t)W * NL + 1;
    size_t half = (n + 1) / 2;
    DBuf b0, b1, wire;
    HIPCHK(b0.alloc(sizeof(i32) * words * half)); HIPCHK(b1.alloc(sizeof(i32) * words * ((half + 1) / 2)));
    // jac: d_pts are the reference's in-memory Jacobian records (144 / 288 bytes, d_inf unused: z == 0 says it).  Many points: level 0 adds
    // them as they are (k_g?_sum0_jac: no inversion anywhere before the one at the end); few: they become wire records for the level programs
    if (jac && half <= g_lat_max) {
        HIPCHK(wire.alloc((size_t)PB * n));
        int rc = jac_to_wire_dev(PB, d_pts, wire.p, nullptr, n, s);
        if (rc) return rc;
        d_pts = wire.as<u8>(); d_inf = nullptr; jac = false;
    }
    if (half <= g_lat_max) {
        // few points: every addition of the tree as a level program, one per wave (k_lat.hip: sum0 / sum1 / sumfin; complete
        // projective formulas, two product levels per addition) -- ~10 us a level instead of ~70-140
        const size_t p0 = W == 3 ? LAT_SUM0_1_OFFSET : LAT_SUM0_2_OFFSET, p1 = W == 3 ? LAT_SUM1_1_OFFSET : LAT_SUM1_2_OFFSET, pf = W == 3 ? LAT_SUMFIN_1_OFFSET : LAT_SUMFIN_2_OFFSET;
        const u8* L = (const u8*)g_gens.lat;
        prof_mark("k_lat:sum");
        hipLaunchKernelGGL(k_lat, dim3((unsigned)half), dim3(64), lat_lds_bytes(p0), s, L + p0, d_pts, (size_t)PB, d_inf, (size_t)0, (const u8*)nullptr, half,
                           (const u8*)nullptr, n, (const u8*)nullptr, (u8*)nullptr, b0.as<u64>(), half);
        i32* src = b0.as<i32>(); i32* dst = b1.as<i32>();
        size_t cur = half;
        while (cur > 1) {
            const size_t h = (cur + 1) / 2;
            hipLaunchKernelGGL(k_lat, dim3((unsigned)h), dim3(64), lat_lds_bytes(p1), s, L + p1, (const u8*)nullptr, (size_t)0, (const u8*)nullptr, (size_t)0,
                               (const u8*)nullptr, h, reinterpret_cast<const u8*>(src), cur, (const u8*)nullptr, (u8*)nullptr, reinterpret_cast<u64*>(dst), h);
            std::swap(src, dst);
            cur = h;
        }
        DBuf good; HIPCHK(good.alloc(1, s));
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), lat_lds_bytes(pf), s, L + pf, (const u8*)nullptr, (size_t)0, (const u8*)nullptr, (size_t)0,
                           (const u8*)nullptr, (size_t)0, reinterpret_cast<const u8*>(src), (size_t)1, (const u8*)nullptr, good.as<u8>(), reinterpret_cast<u64*>(d_out), (size_t)1);
        hipLaunchKernelGGL(k_good_to_flag, dim3(1), dim3(WG), 0, s, (const u8*)good.as<u8>(), d_out_inf);
        prof_mark(nullptr);
        HIPCHK(hipGetLastError());
        if (sync) HIPCHK(hipStreamSynchronize(s));                        // (the temporaries live in the call's arena: an unsynchronised caller orders its own readers)
        return BLSMI_OK;
    }
    prof_mark(W == 3 ? "k_g1_sum0" : "k_g2_sum0");
    if (jac && W == 3) hipLaunchKernelGGL(k_g1_sum0_jac, dim3(nblocks(half)), dim3(WG), 0, s, (const u64*)d_pts, b0.as<i32>(), n, half);
    else if (jac) hipLaunchKernelGGL(k_g2_sum0_jac, dim3(nblocks(half)), dim3(WG), 0, s, (const u64*)d_pts, b0.as<i32>(), n, half);
    else hipLaunchKernelGGL(k0, dim3(nblocks(half)), dim3(WG), 0, s, d_pts, d_inf, b0.as<i32>(), n, half);
    prof_mark(W == 3 ? "k_g1_sum" : "k_g2_sum");
    i32* src = b0.as<i32>(); i32* dst = b1.as<i32>();
    size_t cur = half;
    while (cur > 1) {
        const size_t h = (cur + 1) / 2;
        hipLaunchKernelGGL(k1, dim3(nblocks(h)), dim3(WG), 0, s, (const i32*)src, dst, cur, h);
        std::swap(src, dst);
        cur = h;
    }
    prof_mark(W == 3 ? "k_g1_sum_final" : "k_g2_sum_final");
    hipLaunchKernelGGL(kfinal, dim3(1), dim3(WG), 0, s, (const i32*)src, d_out, d_out_inf);
    prof_mark(nullptr);
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(hipStreamSynchronize(s));
    return BLSMI_OK;
}
template <int PB, int W, class K0, class K1, class K2>
static int sum_host(K0 k0, K1 k1, K2 kfinal, const uint8_t* pts, const uint8_t* in_inf, size_t n, uint8_t* out, int* out_inf, bool jac = false, uint64_t* out_jac = nullptr) {
    // jac: pts are in-memory Jacobian records (no in_inf); out_jac (may be null): the sum as such a record with z = 1, (0, 1, 0) for infinity
    if (!out_inf || (!out && !out_jac) || (n && !pts)) return BLSMI_E_ARG;
    const size_t jb = rec_bytes(PB, true);
    if (n == 0) {                                                          // empty sum = infinity (g2pubs/bls.go:166, 181)
        if (out) memset(out, 0, PB);
        if (out_jac) { static const uint64_t one[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
                       memset(out_jac, 0, jb); memcpy(reinterpret_cast<uint8_t*>(out_jac) + PB / 2, one, 48); }   // FQOne (fq.go:22) in y: G?ProjectiveZero
        *out_inf = 1; return BLSMI_OK;
    }
    LOCK_AND_INIT();
    DBuf dp, di, dout, dflag, dj;
    HIPCHK(dp.alloc((jac ? jb : (size_t)PB) * n)); HIPCHK(di.alloc(n)); HIPCHK(dout.alloc(PB)); HIPCHK(dflag.alloc(sizeof(i32))); HIPCHK(dj.alloc(jb));
    HIPCHK(hipMemcpyAsync(dp.p, pts, (jac ? jb : (size_t)PB) * n, hipMemcpyHostToDevice, g_stream));
    if (in_inf && !jac) HIPCHK(hipMemcpyAsync(di.p, in_inf, n, hipMemcpyHostToDevice, g_stream));
    int rc = sum_dev<PB, W>(k0, k1, kfinal, dp.as<u8>(), (in_inf && !jac) ? di.as<u8>() : nullptr, n, dout.as<u8>(), dflag.as<i32>(), g_stream, true, jac);
    if (rc) return rc;
    i32 flag = 0;
    if (out_jac) {
        hipLaunchKernelGGL(k_affine_to_jac, dim3(1), dim3(WG), 0, g_stream, (const u8*)dout.as<u8>(), (const void*)dflag.p, 0, W == 3 ? 1 : 2, dj.as<u64>(), (size_t)1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out_jac, dj.p, jb, hipMemcpyDeviceToHost, g_stream));
    }
    if (out) HIPCHK(hipMemcpyAsync(out, dout.p, PB, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    HIPCHK(hipMemcpyAsync(&flag, dflag.p, sizeof flag, hipMemcpyDeviceToHost, g_stream)); HIPCHK(hipStreamSynchronize(g_stream));
    *out_inf = flag;
    return BLSMI_OK;
}
BLSMI_API int blsmi_g1_sum(const uint8_t* pts, const uint8_t* in_inf, size_t n, uint8_t out[96], int* out_inf) { return sum_host<96, 3>(k_g1_sum0, k_g1_sum, k_g1_sum_final, pts, in_inf, n, out, out_inf); }
BLSMI_API int blsmi_g2_sum(const uint8_t* pts, const uint8_t* in_inf, size_t n, uint8_t out[192], int* out_inf) { return sum_host<192, 6>(k_g2_sum0, k_g2_sum, k_g2_sum_final, pts, in_inf, n, out, out_inf); }
// AggregateSignatures / AggregatePublicKeys (g2pubs/bls.go:165-192, g1pubs/bls.go:177-204) over the points as the Go values hold them, the sum
// handed back the same way (z = 1): no ToAffine, no SerializeBytes, no FQReprToFQ in the shim
BLSMI_API int blsmi_g1_sum_jac(const uint64_t* pts_jac, size_t n, uint64_t out_jac[18], int* out_inf) {
    return sum_host<96, 3>(k_g1_sum0, k_g1_sum, k_g1_sum_final, reinterpret_cast<const uint8_t*>(pts_jac), nullptr, n, nullptr, out_inf, true, out_jac);
}
BLSMI_API int blsmi_g2_sum_jac(const uint64_t* pts_jac, size_t n, uint64_t out_jac[36], int* out_inf) {
    return sum_host<192, 6>(k_g2_sum0, k_g2_sum, k_g2_sum_final, reinterpret_cast<const uint8_t*>(pts_jac), nullptr, n, nullptr, out_inf, true, out_jac);
}
template <int PB, int W, class K0, class K1, class K2>
static int sum_dev_api(K0 k0, K1 k1, K2 kfinal, const void* d_pts, const void* d_in_inf, size_t n, void* d_out, int* out_inf, void* stream) {
    if (!d_out || !out_inf || (n && !d_pts)) return BLSMI_E_ARG;
    LOCK_AND_INIT_AT(d_out);
    UseStream us(stream);
    if (n == 0) { HIPCHK(hipMemsetAsync(d_out, 0, PB, g_stream)); HIPCHK(hipStreamSynchronize(g_stream)); *out_inf = 1; return BLSMI_OK; }
    DBuf dflag; HIPCHK(dflag.alloc(sizeof(i32)));
    int rc = sum_dev<PB, W>(k0, k1, kfinal, (const u8*)d_pts, (const u8*)d_in_inf, n, (u8*)d_out, dflag.as<i32>(), g_stream);
    if (rc) return rc;
    i32 flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, dflag.p, sizeof flag, hipMemcpyDeviceToHost, g_stream)); HIPCHK(hipStreamSynchronize(g_stream));
    *out_inf = flag;
    return BLSMI_OK;
}
BLSMI_API int blsmi_g1_sum_dev(const void* d_pts, const void* d_in_inf, size_t n, void* d_out, int* out_inf, void* stream) { return sum_dev_api<96, 3>(k_g1_sum0, k_g1_sum, k_g1_sum_final, d_pts, d_in_inf, n, d_out, out_inf, stream); }
BLSMI_API int blsmi_g2_sum_dev(const void* d_pts, const void* d_in_inf, size_t n, void* d_out, int* out_inf, void* stream) { return sum_dev_api<192, 6>(k_g2_sum0, k_g2_sum, k_g2_sum_final, d_pts, d_in_inf, n, d_out, out_inf, stream); }

// multi-scalar multiplication sum_i k_i * P_i.  Small batches: per-point windowed multiples feed the tree sum on the
// device (one launch of latency, ~6 ms up to 64k points).  From BLSMI_MSM_BUCKET_MIN points (default 2^17, where the
// two cross over) on: the bucket method of msm.inc, whose fixed tail (chunk reduction + 240 serial doublings) is ~6 ms.
constexpr int BLSMI_E_SKEW = -1000;    // internal: bucket method declined (unbalanced digits)
struct MsmKernels {
    void (*bucket)(const u8*, const u32*, const u32*, const u32*, const u32*, i32*, size_t, int, size_t);
    void (*chunk)(const i32*, i32*, int, int, size_t, size_t);
    void (*fold)(const i32*, i32*, size_t, size_t, int);
    void (*final)(const i32*, int, int, u8*, i32*);
};
template <int PB, int W>
static int msm_bucket_dev(const MsmKernels& k, const u8* d_pts, const u8* d_scalars, size_t n, u8* d_out, i32* d_flag, hipStream_t s) {
    // 16-bit windows: n / 2^16 points per bucket, and a 255-bit scalar still fills 15 bits of the top window (a window
    // size that leaves the top window a few bits wide would pile every point into a handful of buckets there)
    const int c = 16;
    const int K = 16;                                                      // buckets per chunk lane: 2^12 chunk lanes per window keep every SIMD busy
    const int nwin = (256 + c - 1) / c;
    const size_t B1 = (size_t)1 << c, nb = B1 * nwin, per_win = B1 / K, nct = per_win * nwin;
    const size_t jw = (size_t)W * NL + 1;                                  // words of one Jacobian SoA record
    DBuf hist, offs, cursor, idx, buckets, ch0, ch1, dmax;
    HIPCHK(hist.alloc(sizeof(u32) * nb)); HIPCHK(offs.alloc(sizeof(u32) * nb)); HIPCHK(cursor.alloc(sizeof(u32) * nb)); HIPCHK(dmax.alloc(sizeof(u32)));
    HIPCHK(idx.alloc(sizeof(u32) * n * nwin)); HIPCHK(buckets.alloc(sizeof(i32) * jw * nb));
    HIPCHK(ch0.alloc(sizeof(i32) * jw * nct)); HIPCHK(ch1.alloc(sizeof(i32) * jw * ((per_win + 1) / 2) * nwin));
    HIPCHK(hipMemsetAsync(hist.p, 0, sizeof(u32) * nb, s));
    HIPCHK(hipMemsetAsync(dmax.p, 0, sizeof(u32), s));
    prof_mark("k_msm_hist");
    hipLaunchKernelGGL(k_msm_hist, dim3(nblocks(n)), dim3(WG), 0, s, d_scalars, n, c, nwin, hist.as<u32>());
    prof_mark("k_msm_max");
    // Skewed scalars (many equal digits) would leave one lane adding a whole bucket by itself: beyond 2048 points in
    // any bucket the caller falls back to the per-point multiples, whose cost does not depend on the scalars.
    hipLaunchKernelGGL(k_msm_max, dim3(nblocks(nb)), dim3(WG), 0, s, (const u32*)hist.as<u32>(), nb, dmax.as<u32>());
    u32 biggest = 0;
    HIPCHK(hipMemcpyAsync(&biggest, dmax.p, sizeof biggest, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (biggest > 2048) return BLSMI_E_SKEW;
    prof_mark("k_msm_scan");
    hipLaunchKernelGGL(k_msm_scan, dim3(nwin), dim3(256), 0, s, (const u32*)hist.as<u32>(), offs.as<u32>(), cursor.as<u32>(), c);
    prof_mark("k_msm_scatter");
    hipLaunchKernelGGL(k_msm_scatter, dim3(nblocks(n)), dim3(WG), 0, s, d_scalars, n, c, nwin, cursor.as<u32>(), idx.as<u32>());
    // buckets ranked by population, so that the 64 lanes of a wave add about the same number of points (msm.inc)
    DBuf cls, perm;
    HIPCHK(cls.alloc(sizeof(u32) * 768, s)); HIPCHK(perm.alloc(sizeof(u32) * nb, s));
    HIPCHK(hipMemsetAsync(cls.p, 0, sizeof(u32) * 256, s));
    const unsigned cb = (unsigned)((nb + 255) / 256);
    prof_mark("k_msm_class_*");
    hipLaunchKernelGGL(k_msm_class_hist, dim3(cb), dim3(256), 0, s, (const u32*)hist.as<u32>(), nb, cls.as<u32>());
    hipLaunchKernelGGL(k_msm_class_scan, dim3(1), dim3(256), 0, s, (const u32*)cls.as<u32>(), cls.as<u32>() + 256, cls.as<u32>() + 512);
    hipLaunchKernelGGL(k_msm_class_scatter, dim3(cb), dim3(256), 0, s, (const u32*)hist.as<u32>(), nb, (const u32*)(cls.as<u32>() + 256), cls.as<u32>() + 512, perm.as<u32>());
    prof_mark(W == 6 ? (g_pair_layout ? "k_g2_msm_bucket_pair" : "k_g2_msm_bucket") : "k_g1_msm_bucket");
    if (W == 6 && g_pair_layout)                                           // G2: a lane pair per bucket, two waves per SIMD
        hipLaunchKernelGGL(k_g2_msm_bucket_pair, dim3((unsigned)((nb + PT - 1) / PT)), dim3(WG), 0, s, d_pts, (const u32*)idx.as<u32>(), (const u32*)offs.as<u32>(), (const u32*)hist.as<u32>(), (const u32*)perm.as<u32>(), buckets.as<i32>(), n, c, nb);
    else
        hipLaunchKernelGGL(k.bucket, dim3(nblocks(nb)), dim3(WG), 0, s, d_pts, (const u32*)idx.as<u32>(), (const u32*)offs.as<u32>(), (const u32*)hist.as<u32>(), (const u32*)perm.as<u32>(), buckets.as<i32>(), n, c, nb);
    prof_mark(W == 6 ? "k_g2_msm_chunk" : "k_g1_msm_chunk");
    hipLaunchKernelGGL(k.chunk, dim3(nblocks(nct)), dim3(WG), 0, s, (const i32*)buckets.as<i32>(), ch0.as<i32>(), c, K, nb, nct);
    prof_mark(W == 6 ? "k_g2_msm_fold" : "k_g1_msm_fold");
    i32* src = ch0.as<i32>(); i32* dst = ch1.as<i32>();
    size_t seg = per_win;
    while (seg > 1) {
        const size_t half = (seg + 1) / 2;
        hipLaunchKernelGGL(k.fold, dim3(nblocks(half * nwin)), dim3(WG), 0, s, (const i32*)src, dst, seg, half, nwin);
        std::swap(src, dst);
        seg = half;
    }
    // Horner over the 16 windows (240 dependent doublings) and ToAffine on one lane: this is the path for ARBITRARY curve points
    // (blsmi_set_mul_assume_subgroup(0)); the default path (msm_bucket_glv_dev) has a level program for its shorter tail
    static_assert(W == 3 || W == 6, "G1 / G2");
    prof_mark(W == 6 ? "k_g2_msm_final" : "k_g1_msm_final");
    hipLaunchKernelGGL(k.final, dim3(1), dim3(WG), 0, s, (const i32*)src, nwin, c, d_out, d_flag);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));                                       // temporaries die with this scope
    return BLSMI_OK;
}
// The bucket method for points of the prime-order subgroup (msm.inc, second half): scalars decomposed through the endomorphisms
// (G1: 8 bucket-windows fed by 2 n items, G2: 4 fed by 4 n), points converted to raw limbs once.  Same result as msm_bucket_dev.
template <int PB, int W>
static int msm_bucket_glv_dev(const MsmKernels& k, const u8* d_pts, const u8* d_scalars, size_t n, u8* d_out, i32* d_flag, hipStream_t s,
                              const std::function<int()>& points_arrive = {}) {
    constexpr int c = 16, K = 8;                                           // 8 buckets per chunk lane: 2^13 chunks per window, 13 fold levels (the tail programs are generated for that)
    constexpr int nbw = W == 3 ? 8 : 4, sh = W == 3 ? 3 : 2, NS = 16 / nbw;
    constexpr size_t RAWW = W == 3 ? 48 : 244;
    const size_t B1 = (size_t)1 << c, nb = B1 * nbw, per_win_chunks = B1 / K, nct = per_win_chunks * nbw, per_win_items = (size_t)NS * n;
    const size_t jw = (size_t)W * NL + 1;
    if (n >= ((size_t)1 << 30)) return BLSMI_E_ARG;                         // the item word keeps the stream in bits 30-31
    DBuf raw, rec, hist, offs, cursor, idx, buckets, ch0, ch1, dmax, cls, perm;
    HIPCHK(raw.alloc(sizeof(i32) * RAWW * n, s)); HIPCHK(rec.alloc(32 * n, s));
    HIPCHK(hist.alloc(sizeof(u32) * nb, s)); HIPCHK(offs.alloc(sizeof(u32) * nb, s)); HIPCHK(cursor.alloc(sizeof(u32) * nb, s)); HIPCHK(dmax.alloc(sizeof(u32), s));
    // grouping the 16 n items by bucket: a device radix sort (msm.inc: k_msm_items, k_util.hip); BLSMI_MSM_SORT=0 at start-up or
    // blsmi_set_option("msm_sort", 0) (the tests cross-check the two) takes the exact histogram + scan + atomic scatter instead
    const size_t nitems = (size_t)16 * n;
    bool sort_mode = nitems < ((size_t)1 << 31) && g_msm_sort.load(std::memory_order_relaxed);   // (the sort counts its items in an int: 2^27 points and beyond take the exact passes)
    DBuf skey[2], sval[2];
    if (sort_mode) {
        // the sort's ping-pong buffers are 256 bytes per point (4 x 16 n words) on top of everything else: when they do not fit, the exact passes
        // (64 bytes per point) serve instead of failing the call
        //  ... and what an abandoned attempt did get goes back first: the exact passes' index buffer may need exactly that memory
        const Arena::Mark before = tl_ctx->arena.mark();
        for (int i = 0; i < 2 && sort_mode; i++)
            if (skey[i].alloc(sizeof(u32) * nitems, s) != hipSuccess || sval[i].alloc(sizeof(u32) * nitems, s) != hipSuccess) { (void)hipGetLastError(); sort_mode = false; }
        if (!sort_mode) { tl_ctx->arena.rewind(before); for (int i = 0; i < 2; i++) { skey[i].p = nullptr; sval[i].p = nullptr; } }
    }
    if (!sort_mode) { if (idx.alloc(sizeof(u32) * per_win_items * nbw, s) != hipSuccess) { (void)hipGetLastError(); return BLSMI_E_NOMEM; } }
    HIPCHK(buckets.alloc(sizeof(i32) * jw * nb, s));
    HIPCHK(ch0.alloc(sizeof(i32) * jw * 2 * nct, s)); HIPCHK(ch1.alloc(sizeof(i32) * jw * 2 * nct, s));   // the fold's arrays: at most 2 nct records on either side
    HIPCHK(cls.alloc(sizeof(u32) * 768, s)); HIPCHK(perm.alloc(sizeof(u32) * nb, s));
    HIPCHK(hipMemsetAsync(hist.p, 0, sizeof(u32) * nb, s));
    HIPCHK(hipMemsetAsync(dmax.p, 0, sizeof(u32), s));
    HIPCHK(hipMemsetAsync(cls.p, 0, sizeof(u32) * 256, s));
    prof_mark(W == 3 ? "k_msm_recode_g1" : "k_msm_recode_g2");
    if (W == 3) hipLaunchKernelGGL(k_msm_recode_g1, dim3(nblocks(n)), dim3(WG), 0, s, d_scalars, rec.as<u8>(), n);
    else hipLaunchKernelGGL(k_msm_recode_g2, dim3(nblocks(n)), dim3(WG), 0, s, d_scalars, rec.as<u8>(), n);
    u32 biggest = 0;
    size_t slice_stride = per_win_items;                                   // a bucket's items: idx + (bucket >> 16) * slice_stride + offs[bucket]
    auto exact_passes = [&]() -> int {
        {
            prof_mark("k_msm_hist_glv");
            hipLaunchKernelGGL(k_msm_hist_glv, dim3(nblocks(n)), dim3(WG), 0, s, (const u8*)rec.as<u8>(), n, nbw, hist.as<u32>());
            prof_mark("k_msm_max");
            hipLaunchKernelGGL(k_msm_max, dim3(nblocks(nb)), dim3(WG), 0, s, (const u32*)hist.as<u32>(), nb, dmax.as<u32>());
            HIPCHK(hipMemcpyAsync(&biggest, dmax.p, sizeof biggest, hipMemcpyDeviceToHost, s));
        }
        prof_mark("k_msm_scan");
        hipLaunchKernelGGL(k_msm_scan, dim3(nbw), dim3(256), 0, s, (const u32*)hist.as<u32>(), offs.as<u32>(), cursor.as<u32>(), c);
        prof_mark("k_msm_scatter_glv");
        hipLaunchKernelGGL(k_msm_scatter_glv, dim3(nblocks(n)), dim3(WG), 0, s, (const u8*)rec.as<u8>(), n, nbw, sh, cursor.as<u32>(), idx.as<u32>());
        slice_stride = per_win_items;
        return BLSMI_OK;
    };
    const u32* items = nullptr;                                            // the bucket pass reads items + (bucket >> 16) * slice_stride + offs[bucket]
    if (sort_mode) {
        prof_mark("k_msm_items");
        hipLaunchKernelGGL(k_msm_items, dim3(nblocks(n)), dim3(WG), 0, s, (const u8*)rec.as<u8>(), n, nbw, sh, skey[0].as<u32>(), sval[0].as<u32>());
        prof_mark("rocprim:radix_sort");
        u32* kk[2] = {skey[0].as<u32>(), skey[1].as<u32>()}; u32* vv[2] = {sval[0].as<u32>(), sval[1].as<u32>()};
        u32 *ks = nullptr, *vs = nullptr;
        int bits = 17; while (((size_t)1 << bits) <= nb) bits++;            // buckets 0 .. nb - 1 and the sentinel nb
        if (const int e = blsmi_util::sort_pairs_async(kk, vv, nitems, bits, s, [](size_t b) { return tl_ctx->arena.alloc(b); }, &ks, &vs)) { (void)hipGetLastError(); return e == (int)hipErrorOutOfMemory ? BLSMI_E_NOMEM : BLSMI_E_HIP; }
        prof_mark("k_msm_runs");
        HIPCHK(hipMemsetAsync(offs.p, 0, sizeof(u32) * nb, s)); HIPCHK(hipMemsetAsync(cursor.p, 0, sizeof(u32) * nb, s));
        hipLaunchKernelGGL(k_msm_runs, dim3(nblocks(nitems)), dim3(WG), 0, s, (const u32*)ks, nitems, (u32)nb, offs.as<u32>(), cursor.as<u32>());
        hipLaunchKernelGGL(k_msm_run_lengths, dim3(nblocks(nb)), dim3(WG), 0, s, (const u32*)offs.as<u32>(), (const u32*)cursor.as<u32>(), nb, hist.as<u32>(), dmax.as<u32>());
        HIPCHK(hipMemcpyAsync(&biggest, dmax.p, sizeof biggest, hipMemcpyDeviceToHost, s));
        items = vs; slice_stride = 0;
    } else { const int rc = exact_passes(); if (rc) return rc; }
    prof_mark("k_msm_class_*");
    const unsigned cb = (unsigned)((nb + 255) / 256);
    hipLaunchKernelGGL(k_msm_class_hist, dim3(cb), dim3(256), 0, s, (const u32*)hist.as<u32>(), nb, cls.as<u32>());
    hipLaunchKernelGGL(k_msm_class_scan, dim3(1), dim3(256), 0, s, (const u32*)cls.as<u32>(), cls.as<u32>() + 256, cls.as<u32>() + 512);
    hipLaunchKernelGGL(k_msm_class_scatter, dim3(cb), dim3(256), 0, s, (const u32*)hist.as<u32>(), nb, (const u32*)(cls.as<u32>() + 256), cls.as<u32>() + 512, perm.as<u32>());
    prof_mark(nullptr);
    // The digit passes above need the scalars only: the host form copies the POINTS now, beside them (points_arrive), and only
    // then are the points converted to raw limbs.  Skewed scalars (a bucket beyond 2048 items would be one lane's serial work)
    // fall back to the per-point multiples, whose cost does not depend on the scalars.
    if (points_arrive) { const int rc = points_arrive(); if (rc) return rc; }
    prof_mark(W == 3 ? "k_msm_rawpts_g1" : "k_msm_rawpts_g2");
    if (W == 3) hipLaunchKernelGGL(k_msm_rawpts_g1, dim3(nblocks(n)), dim3(WG), 0, s, d_pts, raw.as<i32>(), n);
    else hipLaunchKernelGGL(k_msm_rawpts_g2, dim3(nblocks(n)), dim3(WG), 0, s, d_pts, raw.as<i32>(), n);
    prof_mark(nullptr);
    HIPCHK(hipStreamSynchronize(s));
    if (biggest > 2048) return BLSMI_E_SKEW;
    if (!sort_mode) items = idx.as<u32>();
    prof_mark(W == 3 ? "k_g1_msm_bucket_raw" : "k_g2_msm_bucket_raw_pair");
    if (W == 3) hipLaunchKernelGGL(k_g1_msm_bucket_raw, dim3(nblocks(nb)), dim3(WG), 0, s, (const i32*)raw.as<i32>(), items, (const u32*)offs.as<u32>(), (const u32*)hist.as<u32>(), (const u32*)perm.as<u32>(), buckets.as<i32>(), slice_stride, nb);
    else hipLaunchKernelGGL(k_g2_msm_bucket_raw_pair, dim3((unsigned)((nb + PT - 1) / PT)), dim3(WG), 0, s, (const i32*)raw.as<i32>(), items, (const u32*)offs.as<u32>(), (const u32*)hist.as<u32>(), (const u32*)perm.as<u32>(), buckets.as<i32>(), slice_stride, nb);
    // running sums per chunk WITHOUT the per-lane multiplication, then the fold that carries the odd-element sums along (msm.inc):
    // one addition deep per level; out come, per window, X, L and O_0 .. O_{m-1}
    const bool pairk = W == 6 && g_pair_layout;                            // G2: a lane pair per chunk / per sum, two waves per SIMD
    prof_mark(W == 6 ? (pairk ? "k_g2_msm_chunk2_pair" : "k_g2_msm_chunk2") : "k_g1_msm_chunk2");
    if (pairk) hipLaunchKernelGGL(k_g2_msm_chunk2_pair, dim3((unsigned)((nct + PT - 1) / PT)), dim3(WG), 0, s, (const i32*)buckets.as<i32>(), ch0.as<i32>(), c, K, nb, nct);
    else if (W == 6) hipLaunchKernelGGL(k_g2_msm_chunk2, dim3(nblocks(nct)), dim3(WG), 0, s, (const i32*)buckets.as<i32>(), ch0.as<i32>(), c, K, nb, nct);
    else hipLaunchKernelGGL(k_g1_msm_chunk2, dim3(nblocks(nct)), dim3(WG), 0, s, (const i32*)buckets.as<i32>(), ch0.as<i32>(), c, K, nb, nct);
    prof_mark(W == 6 ? (pairk ? "k_g2_msm_fold2_pair" : "k_g2_msm_fold2") : "k_g1_msm_fold2");
    i32* src = ch0.as<i32>(); i32* dst = ch1.as<i32>();
    int narr = 2, m = 0;
    const bool lat_tail = g_lat_max > 0 && per_win_chunks == ((size_t)1 << 13) && K == 8;   // the tail runs as a latency program (k_lat.hip), which reads the inter-kernel record form
    for (size_t len = per_win_chunks; len > 1; len /= 2, narr++, m++) {
        const size_t lanes = (size_t)(narr + 1) * nbw * (len / 2);
        const int io = (lat_tail && len == 2) ? 1 : 0;                     // the last level writes what the tail program reads
        if (pairk) hipLaunchKernelGGL(k_g2_msm_fold2_pair, dim3((unsigned)((lanes + PT - 1) / PT)), dim3(WG), 0, s, (const i32*)src, dst, narr, nbw, len, io);
        else if (W == 6) hipLaunchKernelGGL(k_g2_msm_fold2, dim3(nblocks(lanes)), dim3(WG), 0, s, (const i32*)src, dst, narr, nbw, len, io);
        else hipLaunchKernelGGL(k_g1_msm_fold2, dim3(nblocks(lanes)), dim3(WG), 0, s, (const i32*)src, dst, narr, nbw, len, io);
        std::swap(src, dst);
    }
    const size_t nrec = (size_t)(m + 2) * nbw;                             // X, L, O_0 .. O_{m-1} per window
    if (lat_tail) {                              // Horner over the O's and over the 8 / 4 windows + ToAffine: one wave (k_lat.hip: msmfin1 / msmfin2)
        const size_t prog = W == 3 ? LAT_MSMFIN1_OFFSET : LAT_MSMFIN2_OFFSET;
        DBuf good; HIPCHK(good.alloc(1, s));
        prof_mark(W == 3 ? "k_lat:msmfin1" : "k_lat:msmfin2");
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), lat_lds_bytes(prog), s, (const u8*)g_gens.lat + prog, (const u8*)nullptr, (size_t)0,
                           (const u8*)nullptr, (size_t)0, (const u8*)nullptr, (size_t)0, reinterpret_cast<const u8*>(src), nrec,
                           (const u8*)nullptr, good.as<u8>(), reinterpret_cast<u64*>(d_out), (size_t)1);
        hipLaunchKernelGGL(k_good_to_flag, dim3(1), dim3(WG), 0, s, (const u8*)good.as<u8>(), d_flag);
        prof_mark(nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        return BLSMI_OK;
    }
    prof_mark(W == 6 ? "k_g2_msm_final2" : "k_g1_msm_final2");
    if (W == 6) hipLaunchKernelGGL(k_g2_msm_final2, dim3(1), dim3(WG), 0, s, (const i32*)src, nbw, m, 3, c, d_out, d_flag);
    else hipLaunchKernelGGL(k_g1_msm_final2, dim3(1), dim3(WG), 0, s, (const i32*)src, nbw, m, 3, c, d_out, d_flag);
    prof_mark(nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return BLSMI_OK;
}
// sum_i k_i P_i with everything resident on the leased device: bucket method from bucket_min points on (unless the digits
// are skewed), per-point multiples + tree sum below.  Result: affine bytes at d_out, infinity flag (i32) at d_flag.  Synchronises s.
template <int PB, int W, class KM, class K0, class K1, class K2>
static int msm_dev_core(const MsmKernels& mk, KM kmul, K0 k0, K1 k1, K2 kfinal, const u8* d_pts, const u8* d_scalars, size_t n, u8* d_out, i32* d_flag, hipStream_t s,
                        const std::function<int()>& points_arrive = {}) {
    const size_t bucket_min = g_env.msm_bucket_min;
    int rc = BLSMI_E_SKEW;
    bool arrived = !points_arrive;                                         // the host form hands over its copy of the points: issued at the latest before the first kernel that reads them
    auto arrive_once = [&]() -> int { if (arrived) return BLSMI_OK; arrived = true; return points_arrive(); };
    if (n >= bucket_min && mul_subgroup()) rc = msm_bucket_glv_dev<PB, W>(mk, d_pts, d_scalars, n, d_out, d_flag, s, arrive_once);
    else if (n >= bucket_min) { rc = arrive_once(); if (!rc) rc = msm_bucket_dev<PB, W>(mk, d_pts, d_scalars, n, d_out, d_flag, s); }
    if (rc != BLSMI_E_SKEW) return rc;
    rc = arrive_once();
    if (rc) return rc;
    DBuf dm, dinf;
    HIPCHK(dm.alloc((size_t)PB * n, s)); HIPCHK(dinf.alloc(n, s));
    rc = mul_dev_core<PB>(kmul, d_pts, 0, d_scalars, dm.as<u8>(), dinf.as<u8>(), n, s);
    if (rc) return rc;
    return sum_dev<PB, W>(k0, k1, kfinal, dm.as<u8>(), dinf.as<u8>(), n, d_out, d_flag, s);   // synchronises: the temporaries may go
}
template <int PB, int W, class KM, class K0, class K1, class K2>
static int msm_host(const MsmKernels& mk, KM kmul, K0 k0, K1 k1, K2 kfinal, const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out, int* out_inf, bool any_point = false) {
    if (!out || !out_inf || (n && (!pts || !scalars))) return BLSMI_E_ARG;
    if (n == 0) { memset(out, 0, PB); *out_inf = 1; return BLSMI_OK; }
    LOCK_AND_INIT();
    MulAny any_guard(any_point);
    DBuf dp, ds, dout, dflag;
    HIPCHK(dp.alloc((size_t)PB * n)); HIPCHK(ds.alloc(32 * n)); HIPCHK(dout.alloc(PB)); HIPCHK(dflag.alloc(sizeof(i32)));
    // scalars first; the points follow while the digit passes (which need the scalars only) run
    HIPCHK(hipMemcpyAsync(ds.p, scalars, 32 * n, hipMemcpyHostToDevice, g_stream));
    // (on a side stream: a copy from pageable memory issued on the compute stream would wait for the kernels queued there)
    HIPCHK(tl_ctx->ensure_aux());
    hipStream_t st = g_stream, side = tl_ctx->aux[0];
    hipEvent_t arrived = tl_ctx->join[0];
    int rc = msm_dev_core<PB, W>(mk, kmul, k0, k1, kfinal, dp.as<u8>(), ds.as<u8>(), n, dout.as<u8>(), dflag.as<i32>(), g_stream, [&]() -> int {
        HIPCHK(hipMemcpyAsync(dp.p, pts, (size_t)PB * n, hipMemcpyHostToDevice, side));
        HIPCHK(hipEventRecord(arrived, side));
        HIPCHK(hipStreamWaitEvent(st, arrived, 0));
        return BLSMI_OK;
    });
    if (rc) return rc;
    i32 flag = 0;
    HIPCHK(hipMemcpyAsync(out, dout.p, PB, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemcpyAsync(&flag, dflag.p, sizeof flag, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    *out_inf = flag;
    return BLSMI_OK;
}
template <int PB, int W, class KM, class K0, class K1, class K2>
static int msm_dev_api(const MsmKernels& mk, KM kmul, K0 k0, K1 k1, K2 kfinal, const void* d_pts, const void* d_scalars, size_t n, void* d_out, int* out_inf, void* stream, bool any_point = false) {
    if (!d_out || !out_inf || (n && (!d_pts || !d_scalars))) return BLSMI_E_ARG;
    LOCK_AND_INIT_AT(d_out);
    UseStream us(stream);
    MulAny any_guard(any_point);
    if (n == 0) { HIPCHK(hipMemsetAsync(d_out, 0, PB, g_stream)); HIPCHK(hipStreamSynchronize(g_stream)); *out_inf = 1; return BLSMI_OK; }
    DBuf dflag; HIPCHK(dflag.alloc(sizeof(i32)));
    int rc = msm_dev_core<PB, W>(mk, kmul, k0, k1, kfinal, (const u8*)d_pts, (const u8*)d_scalars, n, (u8*)d_out, dflag.as<i32>(), g_stream);
    if (rc) return rc;
    i32 flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, dflag.p, sizeof flag, hipMemcpyDeviceToHost, g_stream)); HIPCHK(hipStreamSynchronize(g_stream));
    *out_inf = flag;
    return BLSMI_OK;
}
static const MsmKernels g_mk1{k_g1_msm_bucket, k_g1_msm_chunk, k_g1_msm_fold, k_g1_msm_final};
static const MsmKernels g_mk2{k_g2_msm_bucket, k_g2_msm_chunk, k_g2_msm_fold, k_g2_msm_final};
BLSMI_API int blsmi_g1_msm(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t out[96], int* out_inf) {
    return msm_host<96, 3>(g_mk1, k_g1_mul, k_g1_sum0, k_g1_sum, k_g1_sum_final, pts, scalars, n, out, out_inf);
}
BLSMI_API int blsmi_g2_msm(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t out[192], int* out_inf) {
    return msm_host<192, 6>(g_mk2, k_g2_mul, k_g2_sum0, k_g2_sum, k_g2_sum_final, pts, scalars, n, out, out_inf);
}
BLSMI_API int blsmi_g1_msm_dev(const void* d_pts, const void* d_scalars, size_t n, void* d_out, int* out_inf, void* stream) {
    return msm_dev_api<96, 3>(g_mk1, k_g1_mul, k_g1_sum0, k_g1_sum, k_g1_sum_final, d_pts, d_scalars, n, d_out, out_inf, stream);
}
BLSMI_API int blsmi_g2_msm_dev(const void* d_pts, const void* d_scalars, size_t n, void* d_out, int* out_inf, void* stream) {
    return msm_dev_api<192, 6>(g_mk2, k_g2_mul, k_g2_sum0, k_g2_sum, k_g2_sum_final, d_pts, d_scalars, n, d_out, out_inf, stream);
}
#define BLSMI_FLAGS_OK(f) (((f) & ~(unsigned)BLSMI_MUL_ANY_POINT) == 0)
BLSMI_API int blsmi_g1_msm_ex(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t out[96], int* out_inf, unsigned flags) {
    if (!BLSMI_FLAGS_OK(flags)) return BLSMI_E_ARG;
    return msm_host<96, 3>(g_mk1, k_g1_mul, k_g1_sum0, k_g1_sum, k_g1_sum_final, pts, scalars, n, out, out_inf, (flags & BLSMI_MUL_ANY_POINT) != 0);
}
BLSMI_API int blsmi_g2_msm_ex(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t out[192], int* out_inf, unsigned flags) {
    if (!BLSMI_FLAGS_OK(flags)) return BLSMI_E_ARG;
    return msm_host<192, 6>(g_mk2, k_g2_mul, k_g2_sum0, k_g2_sum, k_g2_sum_final, pts, scalars, n, out, out_inf, (flags & BLSMI_MUL_ANY_POINT) != 0);
}
BLSMI_API int blsmi_g1_msm_dev_ex(const void* d_pts, const void* d_scalars, size_t n, void* d_out, int* out_inf, void* stream, unsigned flags) {
    if (!BLSMI_FLAGS_OK(flags)) return BLSMI_E_ARG;
    return msm_dev_api<96, 3>(g_mk1, k_g1_mul, k_g1_sum0, k_g1_sum, k_g1_sum_final, d_pts, d_scalars, n, d_out, out_inf, stream, (flags & BLSMI_MUL_ANY_POINT) != 0);
}
BLSMI_API int blsmi_g2_msm_dev_ex(const void* d_pts, const void* d_scalars, size_t n, void* d_out, int* out_inf, void* stream, unsigned flags) {
    if (!BLSMI_FLAGS_OK(flags)) return BLSMI_E_ARG;
    return msm_dev_api<192, 6>(g_mk2, k_g2_mul, k_g2_sum0, k_g2_sum, k_g2_sum_final, d_pts, d_scalars, n, d_out, out_inf, stream, (flags & BLSMI_MUL_ANY_POINT) != 0);
}

#include "verify_host.inc"


