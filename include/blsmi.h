/*
 * blsmi.h -- C ABI of libblsmi.so: MI355X-native batch BLS12-381 engine behind the
 * g1pubs / g2pubs Verify / VerifyAggregate surface of phoreproject/bls.
 *
 * This is the drop-in boundary: the entry points are what a cgo shim in the reference's Go
 * packages would bind (see INTEGRATION.md for the Go side).  Plain pointers and sizes only.
 *
 * Conventions
 *   - Host entry points (no suffix) take HOST pointers, are blocking and re-entrant; the library
 *     owns device memory and streams.  `_dev` entry points take DEVICE pointers (inputs already
 *     resident in HBM) plus a hipStream_t passed as void* (NULL = the library's stream); the work is
 *     enqueued on that stream and the call returns after synchronising it (results are ready).
 *   - Field elements on the wire are 48-byte big-endian normal form, exactly
 *     G1Affine.SerializeBytes / G2Affine.SerializeBytes of the reference
 *     (g1.go:157-167: x||y, 96 B; g2.go:172-186: x.c0||x.c1||y.c0||y.c1, 192 B).
 *   - Fq12 results are 72 little-endian uint64 per element: the reference's in-memory FQ12
 *     (Montgomery form R = 2^384, coefficient order c0.c0.c0, c0.c0.c1, ... c1.c2.c1 --
 *     pairing_test.go:9-20), bit for bit.
 *   - Scalars are 32-byte big-endian (FRRepr.Bytes, frrepr.go:188-195).
 *   - Return value: 0 on success, negative BLSMI_E_* on error.  Where the reference panics
 *     (a point at infinity entering MillerLoop, pairing.go:17-26/54) the library never aborts:
 *     the tuple gets a defined result (verify -> 0 / false, pairing -> 1) -- see each function.
 */
#ifndef BLSMI_H
#define BLSMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLSMI_OK 0
#define BLSMI_E_NODEVICE (-1)   /* no usable HIP device */
#define BLSMI_E_HIP (-2)        /* a HIP runtime call failed */
#define BLSMI_E_ARG (-3)        /* bad argument (null pointer with n > 0, ...) */
#define BLSMI_E_NOMEM (-4)
#define BLSMI_E_RCCL (-5)       /* librccl could not be loaded, or a collective failed (multi-GPU only) */

/* Bind the calling process to ONE device (HIP ordinal >= 0) and create the library's streams and tables.
 * Idempotent; every other host entry point calls it lazily with device 0. */
int blsmi_init(int device);
/* Drive the first `ndev` devices of this node from this one process (ndev <= 0: every visible device).  Large
 * verify / pairing batches are then split by contiguous block over the devices -- one host thread, stream context
 * and memory pool per shard -- so a single call to blsmi_g{1,2}pubs_verify_batch / _verify_aggregate /
 * blsmi_pairing_batch uses all of them; the only exchanges are the pass/fail bitmap (one RCCL all-reduce over
 * xGMI) and, for an n-way VerifyAggregate, the per-device Fq12 partial products (one RCCL all-gather of 720 bytes
 * per device).  Smaller calls go to the least busy device.  Must be the first call (or repeat the same ndev).
 * Environment: BLSMI_SHARDS (logical shards, default ndev; more shards than devices share devices),
 * BLSMI_SHARD_MIN (smallest batch that is split, default 8192), BLSMI_FORCE_RCCL=1 (build the communicator even
 * for one device).  librccl is loaded with dlopen only when ndev > 1 (or forced) -- BLSMI_RCCL_PATH names the file to load when the
 * process' loader path has none (a Go binary; under torch one is already mapped) --: BLSMI_E_RCCL if that fails. */
int blsmi_init_devices(int ndev);
/* TEST HOOK, not a deployment mode.  With BLSMI_DEVICE_ALIAS=0,0[,0,0] in the environment blsmi_init_devices builds that many LOGICAL
 * devices -- each with its own context pool, streams, generator tables and exchange buffer, as on an N-GPU node -- on the physical GPUs
 * the list names (ndev <= 0: as many as the list has), and, because RCCL refuses two ranks on one GPU, runs the two collectives through
 * a host-staged stand-in with the same buffers and results (blsmi_version() then says ALIASED-DEVICES).  It exists so that a one-GPU
 * box executes the multi-device code: shard -> device routing, per-device pools, owner routing of the *_dev entry points.
 * blsmi_debug_alias_own tells the library which logical device "owns" a device-pointer range (on real hardware the HIP runtime
 * answers that; aliased devices share one ordinal); bytes == 0 forgets the range.  BLSMI_E_ARG outside the hook. */
int blsmi_debug_alias_own(const void *d_ptr, size_t bytes, int device_index);
int blsmi_device_count(void);   /* devices in use (0 before initialisation) */
/* Diagnostic: context leases (entry-point calls + shards of split calls) that device `device_index` (0 .. blsmi_device_count() - 1) has
 * served since initialisation; -1 for no such device.  Lets an operator (and the tests) see that work reaches every device. */
long long blsmi_debug_device_leases(int device_index);
int blsmi_shard_count(void);
void blsmi_shutdown(void);
/* "blsmi <ABI version> gfx950 CUs=.. devices=.. shards=..".  ABI history: 0.2 gave blsmi_g{1,2}pubs_aggregate_partial its trailing
 * `int *bad` argument (a caller built against the 5-argument prototype of 0.1 must be rebuilt); 0.3 adds the *_dev forms of
 * mul / sum / msm / verify_aggregate and changes no existing prototype; 0.4 adds the prepared-key entry points.  Check the prefix
 * before binding by hand.  0.5 adds blsmi_trim / blsmi_held_bytes, the *_ex forms of mul / msm (per-call BLSMI_MUL_ANY_POINT),
 * blsmi_prefer_cpu, blsmi_debug_device_leases and the BLSMI_DEVICE_ALIAS test hook; no existing prototype changes.  0.6 adds the *_jac forms
 * (the reference's in-memory Jacobian / Montgomery points at the boundary); no existing prototype changes.  0.7 adds blsmi_set_row_threshold (the lane-row layout for
 * 2 048 .. 8 192 tuples), the "row_side" / "hash_row_min" / "hash_row_max" / "hash_quad_min" / "hash_quad_max" / "hash_oct_min" / "hash_oct_max" / "hash_g1_quad_min" / "hash_g1_quad_max" options and BLSMI_OP_LANE_ROW / BLSMI_OP_ROW_*_STEP / BLSMI_OP_ROW_G2_* / BLSMI_OP_ROW_CLEAR_H2 for blsmi_debug_op; no existing prototype changes. */
const char *blsmi_version(void);

/* Page-locked ("pinned") host memory for the buffers handed to the host entry points below.  Optional: every entry point takes
 * ordinary (pageable) memory, which the HIP runtime stages at ~10 GB/s; from blsmi_host_alloc memory the copies are single DMAs at
 * PCIe rate (65 536 pairings from host buffers: 26 -> 22 ms).  A cgo caller serialises its points straight into such a buffer
 * (INTEGRATION.md 2e).  The memory is visible to every device the library drives. */
/* Device memory held for future calls.  Every call context (BLSMI_STREAMS per device) keeps the temporaries of its last call for reuse
 * -- steady state makes no allocator call at all -- up to a retention cap of BLSMI_ARENA_KEEP_MB (default 4096) per context: what an
 * outsized call needed beyond the cap is returned to the driver when that call ends.  blsmi_trim gives back what IDLE contexts hold
 * beyond keep_bytes_per_context (0: everything) plus the runtime's pool cache, e.g. after a burst of 2^20-point calls or before another
 * library in the process needs the HBM; *freed_bytes (may be NULL) reports how much.  blsmi_held_bytes: the current total. */
int blsmi_trim(size_t keep_bytes_per_context, size_t *freed_bytes);
size_t blsmi_held_bytes(void);
int blsmi_host_alloc(size_t bytes, void **out);
int blsmi_host_free(void *p);

/* ---- pairing (replaces bls.Pairing, pairing.go:132-136; BASELINE config 2) -------------------
 * out[i] = FinalExponentiation(MillerLoop(P_i, Q_i)) for n independent (P_i in G1, Q_i in G2)
 * affine pairs.  Inputs must be finite curve points (the reference panics on infinity). */
int blsmi_pairing_batch(const uint8_t *g1_aff /* n*96 */, const uint8_t *g2_aff /* n*192 */,
                        uint64_t *out_fq12 /* n*72 */, size_t n);
int blsmi_pairing_batch_dev(const void *d_g1_aff, const void *d_g2_aff, void *d_out_fq12, size_t n, void *stream);
/* Per-kernel HIP-event timing of the last blsmi_pairing_batch[_dev] call (used by bench.py for the
 * roofline object): Miller-loop kernel and final-exponentiation kernel durations in milliseconds. */
int blsmi_set_profiling(int on);
/* Latency path: pairing / verify batches of at most `max_tuples` tuples run ONE TUPLE PER WAVE (the pairing spread over
 * 64 lanes, field elements staged in LDS) instead of one per lane pair: ~10x lower latency for the one-tuple-per-call
 * Go API (g2pubs/bls.go:159-162), same results.  Default 8192 (environment BLSMI_LAT_MAX) -- the two paths cross at ~10 000 tuples --; 0 switches it off. */
int blsmi_set_latency_threshold(size_t max_tuples);
/* Mid-size batches: pairing / verify batches above the latency threshold and of at most `max_tuples` tuples run in the LANE-QUAD layout
 * -- four lanes per tuple, 16 384 tuples = one wave on every SIMD of the chip -- instead of the lane-pair layout, which needs 65 536
 * tuples to fill it (16 384 pairings: 11.5 ms there).  Default 16 384 (environment BLSMI_QUAD_MAX); 0 switches the layout off.
 * Same results bit for bit on all paths. */
int blsmi_set_quad_threshold(size_t max_tuples);
/* A few thousand tuples (blsmi 0.7): a LONE pairing / verify call of min_tuples .. max_tuples tuples runs in the LANE-ROW layout -- sixteen lanes (one DPP
 * row) per tuple, 4 096 tuples = one wave on every SIMD of the chip -- instead of one tuple per wave (below) or per lane quad (above).  Default
 * 2 048 .. 8 192 (environment BLSMI_ROW_MIN / BLSMI_ROW_MAX); max_tuples = 0 switches the layout off.  Calls that find other calls in flight on
 * their device keep the quad kernels (see "crowd_quad" below).  Same results bit for bit on all four paths. */
int blsmi_set_row_threshold(size_t min_tuples, size_t max_tuples);
/* When should a lone call stay on the upstream CPU path?  A call with few elements costs the dependent depth of ONE wave walking the
 * whole computation -- about 0.7 ms for a Miller loop, 1.4 ms for a pairing, a signature or a G2 preparation, 2.0 ms for a Verify --
 * whatever n is, up to a few thousand elements.  Where one CPU core needs less than that for the whole call (BLSSign 0.45 ms,
 * G2AffineToPrepared 0.19 ms, a Jacobian addition 6.5 us: bench.py `reference_shapes`), the shim should not cross the boundary.
 * blsmi_prefer_cpu(shape, n) returns 1 in exactly those cases (n * cpu time per operation < device latency of a lone call), else 0:
 * Sign n <= 3, G2 prepare n <= 7, MillerLoop n = 1, point sums n <= 38; never for Pairing / FinalExponentiation / Verify. */
enum { BLSMI_SHAPE_PAIRING = 0, BLSMI_SHAPE_MILLER_LOOP = 1, BLSMI_SHAPE_FINAL_EXP = 2, BLSMI_SHAPE_G2_PREPARE = 3, BLSMI_SHAPE_VERIFY = 4,
       BLSMI_SHAPE_SIGN = 5, BLSMI_SHAPE_VERIFY_DOMAIN = 6, BLSMI_SHAPE_POINT_ADD = 7 };
int blsmi_prefer_cpu(int shape, size_t n);
/* Environment.  Every BLSMI_* variable is read ONCE, when the library initialises (the first entry point, blsmi_init or blsmi_init_devices),
 * never from an entry point afterwards: changing the environment of a running process changes nothing (and getenv racing setenv is undefined
 * behaviour in a threaded host).  Deployment: BLSMI_STREAMS (call contexts per device, 4), BLSMI_SHARDS, BLSMI_SHARD_MIN, BLSMI_FORCE_RCCL,
 * BLSMI_RCCL_PATH (the librccl to dlopen when several devices are driven -- a Go binary has no torch that maps one), BLSMI_ARENA_KEEP_MB,
 * BLSMI_LAT_MAX / BLSMI_QUAD_MAX / BLSMI_QUAD_MIN (layout hand-overs; also blsmi_set_latency_threshold / _quad_threshold), BLSMI_MUL_GENERIC,
 * BLSMI_COMBINE_MAX / _WAIT_US / _INFLIGHT / _DEBUG (merging of concurrent one-tuple Verify calls).  A/B switches between code paths with
 * identical results: BLSMI_LAYOUT, BLSMI_GEN_LINES, BLSMI_HASH_G1_SPLIT, BLSMI_HASH_G2_PAIR, BLSMI_HASH_G2_PAIR_REDO_EVERY, BLSMI_COFAC2_PAIR,
 * BLSMI_SWU_WAVE_MAX, BLSMI_SIG_SIDE_MAX, BLSMI_SIDE_MAX, BLSMI_FIXED_WAVE_MAX, BLSMI_MSM_BUCKET_MIN, and the ones that can ALSO be
 * switched while running (atomically; a call in flight sees the old or the new value), through blsmi_set_option(name, value):
 *   "agg_cofactor_pow" (BLSMI_AGG_COFACTOR_POW, default 1), "msm_sort" (BLSMI_MSM_SORT, default 1), "dup_force_sort" (BLSMI_DUP_FORCE_SORT, 0),
 *   "lat_rolled" (BLSMI_LAT_ROLLED, default 1; 0: small Pairing calls run the straight-line copy of their level program instead of the one
 *   whose squaring runs are loops), "row_side" (BLSMI_ROW_SIDE, default 1: a g1pubs Verify in the row layout runs its signature side beside the hash),
 *   "hash_row_min" / "hash_row_max" (defaults 2048 / 4096; no environment name): HashG2 of that many messages clears its cofactor sixteen lanes per message
 *   (k_hash_g2_front + k_clear_h2_row, 2.9 -> 2.2 ms for 3 072 messages) instead of a lane pair per message; "hash_quad_min" / "hash_quad_max" (4097 / 16384):
 *   four lanes per message (k_clear_h2_quad, 3.0 -> 2.4 ms for 16 384 messages: 16 384 g1pubs verifies 10.1 -> 9.3 ms); max 0: never.
 *   "hash_oct_min" / "hash_oct_max" (2048 / 7168): EIGHT lanes per message (k_clear_h2_oct: the homogeneous formulas' levels are four and six products wide; takes precedence
 *   over the two above where its range covers the count: 4 096 g1pubs verifies 4.45 -> 4.05 ms, 6 144: 6.54 -> 5.94).
 *   "swu_row_max" (4096): the SWU maps of HashG1 / HashG2 of BLSMI_SWU_WAVE_MAX < n <= swu_row_max messages run a row of sixteen lanes per map (k_swu_g?_rows:
 *   k_swu_g1 0.54 -> 0.24 ms up to 2 048 messages) unless the signature side's kernel runs beside the hash; 0: never.
 *   "row_side_g2pubs" (1): a g2pubs Verify in the row layout runs its signature side beside the hash as g1pubs does ("row_side"); "row_side_piece" (0 = one launch): the side
 *   kernel in launches of that many tuples (measured slower: a piece of 2 048 tuples takes the time of a piece of 4 096); "row_side_lds" (0): bytes of unused LDS per
 *   workgroup of that kernel when the call has more than 4 096 tuples (40960 keeps a wave slot of every SIMD free for the hash; measured slower too).
 *   "hash_g1_quad_min" / "hash_g1_quad_max" (1280 / 32768): HashG1 of that many messages runs its tail -- sum, 11-isogeny, cofactor -- four lanes per message
 *   (k_hash_g1_finish_quad: 0.96 -> 0.48 ms; 4 096 g2pubs verifies 4.35 -> 3.78 ms, 16 384: 8.10 -> 7.57 ms).
 * Layout by what the DEVICE carries (blsmi 0.6): the hand-overs above are a lone caller's.  Calls that arrive together share the chip, and
 * under load the quad kernels serve 2.7x the tuples per second of the one-tuple-per-wave path, so a pairing / verify call of at least
 * "crowd_floor" tuples (BLSMI_CROWD_FLOOR, default 1536) takes them when its tuples plus those of the other calls in flight on its device pass
 * BLSMI_QUAD_MIN; "crowd_quad" (BLSMI_CROWD_QUAD, default 1) 0: by the call's own size only.  Same results either way (bit-exact layouts).
 * Concurrent Verify calls of BLSMI_COMBINE_MAX <= n < "combine_mid_max" tuples (BLSMI_COMBINE_MID_MAX, default 8192; 0: never) are merged into one
 * launch among themselves when no call context is free (more callers than BLSMI_STREAMS), like the one-tuple calls below BLSMI_COMBINE_MAX are among
 * theirs.
 * Eight callers x 4 096 g2pubs tuples: 0.73 -> 1.84 M verifies/s with both; four callers x 4 096 pairings: 1.04 -> 2.50 M/s.
 * Concurrency needs hardware queues: the HIP runtime hands its GPU_MAX_HW_QUEUES queues (default 4) to streams in creation order and kernels of one
 * queue run one after the other, so the library creates its call contexts' streams back to back when it initialises (BLSMI_STREAMS = 4 = one queue
 * each).  Supported ceiling (round 6, profiles/r06_queue_matrix.log): BLSMI_STREAMS <= 4 -- the default -- at ANY GPU_MAX_HW_QUEUES (4, 6, 8 measured: 0.73 - 1.18 M verifies/s for 4 - 16 callers).
 * More call contexts than four on more than four hardware queues collapse (0.14 - 0.30 M verifies/s) or abort inside the runtime: HSA_STATUS_ERROR_OUT_OF_RESOURCES -- a hardware queue's scratch is sized
 * for a chip full of waves of the largest private segment it has dispatched (k_cofac2_pair: 7 680 bytes a lane, ~4 GB a queue), and eight such queues are more than the runtime grants
 * (profiles/r06_queue_abort_q8_c8.log).  Callers beyond four merge or wait.
 * Test hooks: BLSMI_DEVICE_ALIAS (above); "assume_load" (tuples pretended to be in flight from other calls).  Unknown option name: BLSMI_E_ARG.  (blsmi 0.6) */
int blsmi_set_option(const char *name, long long value);
int blsmi_last_kernel_ms(float *miller_ms, float *final_exp_ms);
/* General form: with profiling on, every entry point records HIP events on its launch stream between its major kernels.  This
 * returns the calling thread's log since it last asked, as "kernel=ms;kernel=ms;..." in launch order (a name repeats when a
 * kernel is launched several times; "k_lat:<program>" = a level program of the latency path), and clears it.  Return value:
 * the length needed.  Only unsplit calls are logged (the shard threads of a split call keep their own logs). */
int blsmi_last_profile(char *out, size_t cap);
/* Miller loop only (pairing.go:16-75 with one pair per tuple), same output format */
int blsmi_miller_loop_batch(const uint8_t *g1_aff, const uint8_t *g2_aff, uint64_t *out_fq12, size_t n);
/* Final exponentiation only (pairing.go:79-129) on n Fq12 values in the output format */
int blsmi_final_exponentiation_batch(const uint64_t *in_fq12, uint64_t *out_fq12, size_t n);

/* ---- scalar multiplication and point sums (g1.go:80-90, g2.go:92-102, AggregatePublicKeys /
 * AggregateSignatures g2pubs/bls.go:165-192; BASELINE config 3) --------------------------------
 * out[i] = k_i * P_i in affine form; out_inf[i] = 1 when the result is the point at infinity
 * (its 96/192 bytes are then zero).  in_inf may be NULL (all finite). */
/* The multiplicands must be points of the PRIME-ORDER SUBGROUP -- which every point the g1pubs / g2pubs API can hand to a
 * multiplication is (hash points: Sign; the generators: PrivToPub; keys and signatures that passed Deserialize*'s subgroup
 * check).  For them the multiplication runs through the curve endomorphisms (G1: k = k1 + k2 z^2 with phi; G2: base-|x| digits
 * with psi -- bls_amd/csrc/glv.cuh), half / a quarter of the reference's doublings, the same group element and affine bytes.
 * For ARBITRARY curve points (the reference's bit-serial MulFR accepts any; an on-curve point outside the subgroup fed to the default
 * ladder yields a DIFFERENT point, silently) use the *_ex forms below with BLSMI_MUL_ANY_POINT: the choice is made per call and costs
 * nothing to concurrent callers.  blsmi_set_mul_assume_subgroup(0) / BLSMI_MUL_GENERIC=1 change the PROCESS-WIDE default of the plain
 * entry points instead: every call in flight reads it, so set it once before the first call (legacy switch; prefer the flag). */
#define BLSMI_MUL_ANY_POINT 1u   /* multiplicands may lie outside the prime-order subgroup: plain fixed-window ladder / plain bucket MSM */
int blsmi_set_mul_assume_subgroup(int on);
int blsmi_g1_mul_batch(const uint8_t *pts /* n*96 */, const uint8_t *scalars /* n*32 */, uint8_t *out /* n*96 */, uint8_t *out_inf /* n */, size_t n);
int blsmi_g2_mul_batch(const uint8_t *pts /* n*192 */, const uint8_t *scalars /* n*32 */, uint8_t *out /* n*192 */, uint8_t *out_inf /* n */, size_t n);
/* k_i * generator (PrivToPub g2pubs/bls.go:138-140 uses G2, g1pubs/bls.go:144-146 uses G1).
 * NOT side-channel hardened: the scalar-multiplication kernels index a per-lane table by scalar nibbles and take
 * data-dependent paths in the group law (the reference's bit-serial Mul is not constant-time either); the small-batch
 * level program has a fixed instruction sequence and complete group formulas but still addresses an LDS table by the
 * scalar's digits.  They exist to
 * generate and check test/bench inputs and for public scalars; secret keys belong on the upstream pure-Go module
 * (the one-tuple Sign / PrivToPub stay there in the Go shim). */
int blsmi_g1_mul_generator_batch(const uint8_t *scalars /* n*32 */, uint8_t *out /* n*96 */, uint8_t *out_inf /* n */, size_t n);
int blsmi_g2_mul_generator_batch(const uint8_t *scalars /* n*32 */, uint8_t *out /* n*192 */, uint8_t *out_inf /* n */, size_t n);
/* sum of n points (tree reduction on the device; equals the reference's sequential Jacobian sum
 * after ToAffine).  *out_inf = 1 for the point at infinity. */
int blsmi_g1_sum(const uint8_t *pts, const uint8_t *in_inf, size_t n, uint8_t out[96], int *out_inf);
int blsmi_g2_sum(const uint8_t *pts, const uint8_t *in_inf, size_t n, uint8_t out[192], int *out_inf);
/* multi-scalar multiplication sum_i k_i * P_i (BASELINE config 3; equals summing the reference's MulFR results with
 * AddAssign, compared after ToAffine).  Below 2^17 points: fixed-window multiples + tree sum in one device pass;
 * from there on the bucket method (16-bit windows), with a fallback to the former for degenerate scalar sets. */
int blsmi_g1_msm(const uint8_t *pts /* n*96 */, const uint8_t *scalars /* n*32 */, size_t n, uint8_t out[96], int *out_inf);
int blsmi_g2_msm(const uint8_t *pts /* n*192 */, const uint8_t *scalars /* n*32 */, size_t n, uint8_t out[192], int *out_inf);
/* The same four with the ladder chosen PER CALL: flags = 0 (subgroup points, as above) or BLSMI_MUL_ANY_POINT; other bits: BLSMI_E_ARG. */
int blsmi_g1_mul_batch_ex(const uint8_t *pts, const uint8_t *scalars, uint8_t *out, uint8_t *out_inf, size_t n, unsigned flags);
int blsmi_g2_mul_batch_ex(const uint8_t *pts, const uint8_t *scalars, uint8_t *out, uint8_t *out_inf, size_t n, unsigned flags);
int blsmi_g1_msm_ex(const uint8_t *pts, const uint8_t *scalars, size_t n, uint8_t out[96], int *out_inf, unsigned flags);
int blsmi_g2_msm_ex(const uint8_t *pts, const uint8_t *scalars, size_t n, uint8_t out[192], int *out_inf, unsigned flags);
/* Device-pointer forms of the above (BASELINE config 3 with inputs resident in HBM): every d_* buffer lives on ONE of the
 * library's devices (the call runs on the device that owns d_out), layouts as in the host forms; d_pts == NULL multiplies
 * the group generator; d_out_inf is n bytes, d_in_inf may be NULL.  The single-point results of sum / msm stay on the
 * device (96 / 192 bytes at d_out), their infinity flag comes back through the host int.  `stream` as in
 * blsmi_pairing_batch_dev.  (Added in blsmi 0.3.) */
int blsmi_g1_mul_batch_dev(const void *d_pts, const void *d_scalars, void *d_out, void *d_out_inf, size_t n, void *stream);
int blsmi_g2_mul_batch_dev(const void *d_pts, const void *d_scalars, void *d_out, void *d_out_inf, size_t n, void *stream);
int blsmi_g1_sum_dev(const void *d_pts, const void *d_in_inf, size_t n, void *d_out, int *out_inf, void *stream);
int blsmi_g2_sum_dev(const void *d_pts, const void *d_in_inf, size_t n, void *d_out, int *out_inf, void *stream);
int blsmi_g1_msm_dev(const void *d_pts, const void *d_scalars, size_t n, void *d_out, int *out_inf, void *stream);
int blsmi_g2_msm_dev(const void *d_pts, const void *d_scalars, size_t n, void *d_out, int *out_inf, void *stream);
/* ... and with the per-call ladder choice (flags as for blsmi_g1_mul_batch_ex; d_pts must not be NULL) */
int blsmi_g1_mul_batch_dev_ex(const void *d_pts, const void *d_scalars, void *d_out, void *d_out_inf, size_t n, void *stream, unsigned flags);
int blsmi_g2_mul_batch_dev_ex(const void *d_pts, const void *d_scalars, void *d_out, void *d_out_inf, size_t n, void *stream, unsigned flags);
int blsmi_g1_msm_dev_ex(const void *d_pts, const void *d_scalars, size_t n, void *d_out, int *out_inf, void *stream, unsigned flags);
int blsmi_g2_msm_dev_ex(const void *d_pts, const void *d_scalars, size_t n, void *d_out, int *out_inf, void *stream, unsigned flags);

/* ---- hash to curve (HashG1 hash.go:326-331, HashG2 hash.go:405-411, HashG2WithDomain
 * g2.go:1041-1085) -- messages are concatenated in `msgs`, message i = msgs[off[i] .. off[i+1]) -- */
int blsmi_hash_g1_batch(const uint8_t *msgs, const uint64_t *off /* n+1 */, uint8_t *out /* n*96 */, size_t n);
int blsmi_hash_g2_batch(const uint8_t *msgs, const uint64_t *off /* n+1 */, uint8_t *out /* n*192 */, size_t n);
int blsmi_hash_g2_with_domain_batch(const uint8_t *msgs32 /* n*32 */, const uint8_t domain[8], uint8_t *out /* n*192 */, size_t n);

/* ---- Sign for n (message, secret key) pairs in one call: sig_i = sk_i * H(msg_i) (g2pubs/bls.go:132-135; g1pubs/bls.go:132-135,
 * SignWithDomain :138-141): the hash points stay on the device between the two steps.  sks: n*32 bytes, big-endian scalars (FR as
 * the Go API serialises it); out_inf[i] = 1 when sk_i = 0 mod r (the signature is the point at infinity, written as the all-zero
 * record).  Like the scalar multiplications above these are NOT side-channel hardened, and the secret scalars cross the PCIe bus and
 * live in a device temporary for the duration of the call: bulk signing of test / bench inputs and of keys that may leave the host
 * (the shim's SignBatch); a deployment that must keep its keys in host memory stays on the upstream Sign, and
 * blsmi_prefer_cpu(BLSMI_SHAPE_SIGN, n) says when a small call is faster there anyway. */
int blsmi_g2pubs_sign_batch(const uint8_t *msgs, const uint64_t *off /* n+1 */, const uint8_t *sks /* n*32 */, uint8_t *out_sigs /* n*96 */, uint8_t *out_inf /* n */, size_t n);
int blsmi_g1pubs_sign_batch(const uint8_t *msgs, const uint64_t *off /* n+1 */, const uint8_t *sks /* n*32 */, uint8_t *out_sigs /* n*192 */, uint8_t *out_inf /* n */, size_t n);
int blsmi_g1pubs_sign_with_domain_batch(const uint8_t *msgs32 /* n*32 */, const uint8_t domain[8], const uint8_t *sks /* n*32 */, uint8_t *out_sigs /* n*192 */, uint8_t *out_inf /* n */, size_t n);

/* ---- g2pubs: PublicKey in G2 (192 B affine), Signature in G1 (96 B affine), H: msg -> G1 ------
 * verify_batch: ok[i] = g2pubs.Verify(msg_i, pk_i, sig_i) (g2pubs/bls.go:159-162) as one byte per
 * tuple (0/1) and, if ok_bitmap != NULL, packed LSB-first into ceil(n/8) bytes.
 * inf_flags (may be NULL): bit0 = pk_i is infinity, bit1 = sig_i is infinity -> ok[i] = 0. */
int blsmi_g2pubs_verify_batch(const uint8_t *msgs, const uint64_t *off, const uint8_t *pks /* n*192 */, const uint8_t *sigs /* n*96 */,
                              const uint8_t *inf_flags, uint8_t *ok /* n, may be NULL */, uint8_t *ok_bitmap /* ceil(n/8), may be NULL */, size_t n);
/* (*Signature).VerifyAggregate (g2pubs/bls.go:240-270): one aggregate signature over n (pk_i, msg_i),
 * including the duplicate-message rejection.  *ok = 0/1. */
int blsmi_g2pubs_verify_aggregate(const uint8_t *msgs, const uint64_t *off, const uint8_t *pks, const uint8_t sig[96], size_t n, int *ok);
/* (*Signature).VerifyAggregateCommon (g2pubs/bls.go:275-278) */
int blsmi_g2pubs_verify_aggregate_common(const uint8_t *msg, size_t msg_len, const uint8_t *pks, const uint8_t sig[96], size_t n, int *ok);

/* ---- g1pubs: PublicKey in G1 (96 B), Signature in G2 (192 B), H: msg -> G2 (g1pubs/bls.go) ---- */
int blsmi_g1pubs_verify_batch(const uint8_t *msgs, const uint64_t *off, const uint8_t *pks /* n*96 */, const uint8_t *sigs /* n*192 */,
                              const uint8_t *inf_flags, uint8_t *ok, uint8_t *ok_bitmap, size_t n);
int blsmi_g1pubs_verify_aggregate(const uint8_t *msgs, const uint64_t *off, const uint8_t *pks, const uint8_t sig[192], size_t n, int *ok);
int blsmi_g1pubs_verify_aggregate_common(const uint8_t *msg, size_t msg_len, const uint8_t *pks, const uint8_t sig[192], size_t n, int *ok);
/* the *WithDomain family (g1pubs/bls.go:171-174, 294-311): 32-byte messages, 8-byte domain */
int blsmi_g1pubs_verify_with_domain_batch(const uint8_t *msgs32, const uint8_t domain[8], const uint8_t *pks, const uint8_t *sigs,
                                          const uint8_t *inf_flags, uint8_t *ok, uint8_t *ok_bitmap, size_t n);
int blsmi_g1pubs_verify_aggregate_with_domain(const uint8_t *msgs32, const uint8_t domain[8], const uint8_t *pks, const uint8_t sig[192], size_t n, int *ok);
int blsmi_g1pubs_verify_aggregate_common_with_domain(const uint8_t msg32[32], const uint8_t domain[8], const uint8_t *pks, const uint8_t sig[192], size_t n, int *ok);

/* ---- prepared public keys (g2pubs; added in blsmi 0.4) ---------------------------------------------------------------
 * bls.MillerLoop takes its G2 arguments PREPARED (MillerLoopItem{P *G1Affine, Q *G2Prepared}, pairing.go:4-14;
 * G2AffineToPrepared, g2.go:639-801: the 68 line-coefficient triples that depend on Q alone), and g2pubs.Verify prepares the
 * same public key again on every call (CompareTwoPairings, pairing.go:140-147).  A verifier that meets the same keys again and
 * again prepares them ONCE into device memory -- BLSMI_G2_PREPARED_BYTES per key, one table after the other -- and hands the
 * tables to the *_prepared_dev entry points, whose Miller loops read a key's lines instead of recomputing them.
 * d_key_idx: n x uint32, tuple t uses table d_key_idx[t]; NULL: tuple t uses table t.  Indices are NOT range-checked (the tables
 * are plain memory the caller sized): an index beyond the prepared keys reads past them.  A key given as the all-zero record
 * (the point at infinity) keeps that mark in its table: verdict 0, as in the unprepared forms.  Every result is identical to the
 * unprepared entry point on the same keys.  All buffers on ONE of the library's devices; `stream` as in blsmi_pairing_batch_dev. */
#define BLSMI_G2_PREPARED_BYTES 24704
int blsmi_g2_prepare_batch_dev(const void *d_g2_aff /* n*192 */, size_t n, void *d_prepared /* n*BLSMI_G2_PREPARED_BYTES */, void *stream);
/* the reference's G2Prepared.coeffs of n prepared keys: per key 68 x [3]FQ2, each FQ2 as c0 | c1, 6 x uint64 Montgomery limbs each */
int blsmi_g2_prepared_export_dev(const void *d_prepared, size_t n, void *d_out /* n*68*3*12 uint64 */, void *stream);
/* G2AffineToPrepared of n host points straight into that representation (BenchmarkG2Prepare, pairing_test.go:60-81) */
int blsmi_g2_prepare_batch(const uint8_t *g2_aff /* n*192 */, size_t n, uint64_t *out_coeffs /* n*68*3*12 */);
/* For callers without a HIP runtime of their own (the cgo shim): prepare n HOST keys into tables the library allocates on one of
 * its devices; *handle is the device pointer to pass as d_prepared.  _destroy waits for that device to go idle, then frees. */
int blsmi_g2_prepared_create(const uint8_t *g2_aff /* n*192 */, size_t n, void **handle);
int blsmi_g2_prepared_destroy(void *handle);
/* n independent g2pubs.Verify()s with messages, signatures, key indices and results in HOST memory and the keys prepared;
 * arguments as blsmi_g2pubs_verify_batch with (d_prepared, key_idx) in the place of pks.  Runs on the device that holds the
 * tables (not split over devices). */
int blsmi_g2pubs_verify_batch_prepared(const uint8_t *msgs, const uint64_t *msg_off, const void *d_prepared, const uint32_t *key_idx /* n, may be NULL */,
                                       const uint8_t *sigs, const uint8_t *inf_flags, uint8_t *ok, uint8_t *ok_bitmap, size_t n);
/* Pairing(P_t, Q_t) with Q_t prepared; output as blsmi_pairing_batch_dev */
int blsmi_pairing_batch_prepared_dev(const void *d_g1_aff, const void *d_prepared, const void *d_key_idx, void *d_out_fq12, size_t n, void *stream);
/* g2pubs.Verify x n / Signature.VerifyAggregate with prepared public keys; otherwise as the *_dev forms below */
int blsmi_g2pubs_verify_batch_prepared_dev(const void *d_msgs, const void *d_off, const void *d_prepared, const void *d_key_idx,
                                           const void *d_sigs, const void *d_inf_flags, void *d_ok, size_t n, void *stream);
int blsmi_g2pubs_verify_aggregate_prepared(const uint8_t *msgs, const uint64_t *msg_off, const void *d_prepared, const uint32_t *key_idx /* n, may be NULL */,
                                           const uint8_t sig[96], size_t n, int *ok);   /* messages, indices and signature in HOST memory */
int blsmi_g2pubs_verify_aggregate_prepared_dev(const void *d_msgs, const void *d_off, const void *d_prepared, const void *d_key_idx,
                                               const uint8_t sig[96], size_t n, int *ok, void *stream);

/* ---- the reference's in-memory points at the boundary (added in blsmi 0.6) ---------------------------------------------------
 * The Go types hold their points in Jacobian coordinates and Montgomery limbs: Signature{s *bls.G1Projective}, PublicKey{p *bls.G2Projective}
 * (g2pubs/bls.go:13-15, 53-55; swapped in g1pubs), G1Projective{x, y, z FQ} (g1.go:252-256), G2Projective{x, y, z FQ2} (g2.go:298-302),
 * FQ2{c0, c1 FQ} (fq2.go:14-17), FQ{n FQRepr} (fq.go:11-13), FQRepr [6]uint64 little-endian (fqrepr.go:14), Montgomery form R = 2^384.  The
 * affine entry points above make the shim run ToAffine() -- one Fq inversion per point, g1.go:322-340 -- and SerializeBytes() -- two MontReduce
 * and a byte swap, g1.go:157-167 -- on a host core for every point: ~12 us each, 30-70x the device's cost of the whole verify.  The *_jac
 * forms take the struct bytes AS THEY LIE:
 *     G1 point = 18 x uint64 (x | y | z),            BLSMI_G1_JAC_BYTES = 144
 *     G2 point = 36 x uint64 (x.c0 | x.c1 | y.c0 | y.c1 | z.c0 | z.c1), BLSMI_G2_JAC_BYTES = 288
 * i.e. *(*[18]uint64)(unsafe.Pointer(sig.s)) -- one memcpy per point in the shim, nothing else -- and run ToAffine on the device (a wave
 * whose 64 points all have z == 1, what Deserialize* leaves, skips the inversion like g2.go:368).  z == 0 is the point at infinity
 * (G1Projective.IsZero g1.go:287-289): such a tuple gets verdict 0, so there are no inf_flags here.  The reference's arithmetic keeps every
 * FQ below q (fq.go:37-45); a limb image that is NOT below q cannot come out of it and is read as 0, the way FQReprToFQ reads an invalid
 * repr (fq.go:49-56).  Results are identical, bit for bit, to the affine entry points on ToAffine().SerializeBytes() of the same points.
 * Arguments other than the points are as in the affine forms. */
#define BLSMI_G1_JAC_BYTES 144
#define BLSMI_G2_JAC_BYTES 288
/* G?Projective.ToAffine().SerializeBytes() for n points: wire records (all zero for infinity), out_inf[i] = 1 for infinity (may be NULL) */
int blsmi_g1_jac_to_affine_batch(const uint64_t *g1_jac /* n*18 */, uint8_t *out /* n*96 */, uint8_t *out_inf /* n */, size_t n);
int blsmi_g2_jac_to_affine_batch(const uint64_t *g2_jac /* n*36 */, uint8_t *out /* n*192 */, uint8_t *out_inf /* n */, size_t n);
/* bls.Pairing(p *G1Projective, q *G2Projective) (pairing.go:132-136) for n pairs; output as blsmi_pairing_batch */
int blsmi_pairing_batch_jac(const uint64_t *g1_jac /* n*18 */, const uint64_t *g2_jac /* n*36 */, uint64_t *out_fq12 /* n*72 */, size_t n);
/* AggregateSignatures / AggregatePublicKeys (g2pubs/bls.go:165-192): the sum of n in-memory points, handed back as an in-memory point with
 * z = 1 -- (0, 1, 0), the reference's G?ProjectiveZero, for the point at infinity (*out_inf = 1; also for n = 0).  The points are added
 * as they are (Jacobian + Jacobian, g1.go:400-470): no inversion but the one at the end. */
int blsmi_g1_sum_jac(const uint64_t *g1_jac /* n*18 */, size_t n, uint64_t out_jac[18], int *out_inf);
int blsmi_g2_sum_jac(const uint64_t *g2_jac /* n*36 */, size_t n, uint64_t out_jac[36], int *out_inf);
/* Results in the in-memory form: PrivToPub (k_i * generator) and Sign (sk_i * H(m_i)) for n scalars, each result a bls.G?Projective record with
 * z = 1 -- (0, 1, 0) when the scalar is 0 mod r -- so that the shim builds its PublicKey / Signature by one copy (upstream: FQReprToFQ per coordinate).
 * Same caveats as the affine forms (not side-channel hardened; secret scalars cross the bus). */
int blsmi_g1_mul_generator_batch_jac(const uint8_t *scalars /* n*32 */, uint64_t *out_jac /* n*18 */, size_t n);
int blsmi_g2_mul_generator_batch_jac(const uint8_t *scalars /* n*32 */, uint64_t *out_jac /* n*36 */, size_t n);
int blsmi_g2pubs_sign_batch_jac(const uint8_t *msgs, const uint64_t *off /* n+1 */, const uint8_t *sks /* n*32 */, uint64_t *out_sigs_jac /* n*18 */, size_t n);
int blsmi_g1pubs_sign_batch_jac(const uint8_t *msgs, const uint64_t *off /* n+1 */, const uint8_t *sks /* n*32 */, uint64_t *out_sigs_jac /* n*36 */, size_t n);
int blsmi_g1pubs_sign_with_domain_batch_jac(const uint8_t *msgs32 /* n*32 */, const uint8_t domain[8], const uint8_t *sks /* n*32 */, uint64_t *out_sigs_jac /* n*36 */, size_t n);
/* g2pubs (PublicKey = G2Projective 36 x u64, Signature = G1Projective 18 x u64) */
int blsmi_g2pubs_verify_batch_jac(const uint8_t *msgs, const uint64_t *off, const uint64_t *pks /* n*36 */, const uint64_t *sigs /* n*18 */,
                                  uint8_t *ok /* n, may be NULL */, uint8_t *ok_bitmap /* ceil(n/8), may be NULL */, size_t n);
int blsmi_g2pubs_verify_aggregate_jac(const uint8_t *msgs, const uint64_t *off, const uint64_t *pks /* n*36 */, const uint64_t sig[18], size_t n, int *ok);
int blsmi_g2pubs_verify_aggregate_common_jac(const uint8_t *msg, size_t msg_len, const uint64_t *pks /* n*36 */, const uint64_t sig[18], size_t n, int *ok);
/* g1pubs (PublicKey = G1Projective 18 x u64, Signature = G2Projective 36 x u64) */
int blsmi_g1pubs_verify_batch_jac(const uint8_t *msgs, const uint64_t *off, const uint64_t *pks /* n*18 */, const uint64_t *sigs /* n*36 */,
                                  uint8_t *ok, uint8_t *ok_bitmap, size_t n);
int blsmi_g1pubs_verify_with_domain_batch_jac(const uint8_t *msgs32, const uint8_t domain[8], const uint64_t *pks /* n*18 */, const uint64_t *sigs /* n*36 */,
                                              uint8_t *ok, uint8_t *ok_bitmap, size_t n);
int blsmi_g1pubs_verify_aggregate_jac(const uint8_t *msgs, const uint64_t *off, const uint64_t *pks /* n*18 */, const uint64_t sig[36], size_t n, int *ok);
int blsmi_g1pubs_verify_aggregate_with_domain_jac(const uint8_t *msgs32, const uint8_t domain[8], const uint64_t *pks /* n*18 */, const uint64_t sig[36], size_t n, int *ok);
int blsmi_g1pubs_verify_aggregate_common_jac(const uint8_t *msg, size_t msg_len, const uint64_t *pks /* n*18 */, const uint64_t sig[36], size_t n, int *ok);
int blsmi_g1pubs_verify_aggregate_common_with_domain_jac(const uint8_t msg32[32], const uint8_t domain[8], const uint64_t *pks /* n*18 */, const uint64_t sig[36], size_t n, int *ok);
/* device-pointer forms (every buffer on ONE of the library's devices; `stream` as in blsmi_pairing_batch_dev): the points resident in HBM as the
 * Go side holds them.  d_ok: n verdict bytes on the device. */
int blsmi_pairing_batch_jac_dev(const void *d_g1_jac, const void *d_g2_jac, void *d_out_fq12, size_t n, void *stream);
int blsmi_g2pubs_verify_batch_jac_dev(const void *d_msgs, const void *d_off, const void *d_pks_jac, const void *d_sigs_jac, void *d_ok, size_t n, void *stream);
int blsmi_g1pubs_verify_batch_jac_dev(const void *d_msgs, const void *d_off, const void *d_pks_jac, const void *d_sigs_jac, void *d_ok, size_t n, void *stream);
int blsmi_g1pubs_verify_with_domain_batch_jac_dev(const void *d_msgs32, const void *d_domain8, const void *d_pks_jac, const void *d_sigs_jac, void *d_ok, size_t n, void *stream);
/* prepared keys (see above) made from, and verified against, in-memory points */
int blsmi_g2_prepared_create_jac(const uint64_t *g2_jac /* n*36 */, size_t n, void **handle);
int blsmi_g2pubs_verify_batch_prepared_jac(const uint8_t *msgs, const uint64_t *msg_off, const void *d_prepared, const uint32_t *key_idx /* n, may be NULL */,
                                           const uint64_t *sigs /* n*18 */, uint8_t *ok, uint8_t *ok_bitmap, size_t n);
int blsmi_g2pubs_verify_aggregate_prepared_jac(const uint8_t *msgs, const uint64_t *msg_off, const void *d_prepared, const uint32_t *key_idx /* n, may be NULL */,
                                               const uint64_t sig[18], size_t n, int *ok);

/* Multi-GPU VerifyAggregate: each rank computes the product of its shard's Miller loops
 * prod_i ML(H(m_i), pk_i) (no final exponentiation) as one Fq12 in the wire format; the ranks all-gather
 * the 576-byte partials, multiply them (blsmi_fq12_product) and finish with one final exponentiation.  The value is
 * meaningful only under that final exponentiation: it is a Miller value of the product, not a fixed representative
 * (projective line functions; from 65 536 messages a g2pubs shard pairs its hash points before their cofactor clearing
 * and raises its product to the cofactor multiplier instead). */
int blsmi_g2pubs_aggregate_partial(const uint8_t *msgs, const uint64_t *off, const uint8_t *pks, size_t n, uint64_t *out_fq12 /* 72 */, int *bad /* may be NULL: 1 if a key is infinity */);
int blsmi_g1pubs_aggregate_partial(const uint8_t *msgs, const uint64_t *off, const uint8_t *pks, size_t n, uint64_t *out_fq12 /* 72 */, int *bad);
int blsmi_fq12_product(const uint64_t *in_fq12 /* n*72 */, size_t n, uint64_t *out_fq12 /* 72 */);

/* device-pointer forms of the verify batches (inputs resident in HBM; ok is n bytes on the device) */
int blsmi_g2pubs_verify_batch_dev(const void *d_msgs, const void *d_off, const void *d_pks, const void *d_sigs, const void *d_inf_flags, void *d_ok, size_t n, void *stream);
int blsmi_g1pubs_verify_batch_dev(const void *d_msgs, const void *d_off, const void *d_pks, const void *d_sigs, const void *d_inf_flags, void *d_ok, size_t n, void *stream);
/* VerifyWithDomain (g1pubs/bls.go:171-174): n 32-byte messages and the 8-byte domain resident on the device */
int blsmi_g1pubs_verify_with_domain_batch_dev(const void *d_msgs32, const void *d_domain8, const void *d_pks, const void *d_sigs, const void *d_inf_flags, void *d_ok, size_t n, void *stream);

/* device-pointer forms of VerifyAggregate (g2pubs/bls.go:240-270, g1pubs/bls.go:252-282, :300-311): messages, offsets (or the
 * 8-byte domain) and keys resident on one of the library's devices, the aggregate signature in HOST memory (one point).  The
 * duplicate-message rejection runs on the device as well (a keyed open-addressing table; every occupied slot of a probe sequence is
 * compared exactly).
 * The call runs on the device that owns d_pks and is not split over devices.  (Added in blsmi 0.3.) */
int blsmi_g2pubs_verify_aggregate_dev(const void *d_msgs, const void *d_off, const void *d_pks, const uint8_t sig[96], size_t n, int *ok, void *stream);
int blsmi_g1pubs_verify_aggregate_dev(const void *d_msgs, const void *d_off, const void *d_pks, const uint8_t sig[192], size_t n, int *ok, void *stream);
int blsmi_g1pubs_verify_aggregate_with_domain_dev(const void *d_msgs32, const void *d_domain, const void *d_pks, const uint8_t sig[192], size_t n, int *ok, void *stream);

/* device-pointer forms of VerifyAggregateCommon (g2pubs/bls.go:275-278, g1pubs/bls.go:287-297): the n public keys resident on one of
 * the library's devices (a validator set kept in HBM) are summed where they lie, then one Verify runs; message, domain and signature are
 * HOST bytes.  n = 0: the empty sum is the point at infinity, verdict 0.  (blsmi 0.5) */
int blsmi_g2pubs_verify_aggregate_common_dev(const void *d_pks /* n*192 */, size_t n, const uint8_t *msg, size_t msg_len, const uint8_t sig[96], int *ok, void *stream);
int blsmi_g1pubs_verify_aggregate_common_dev(const void *d_pks /* n*96 */, size_t n, const uint8_t *msg, size_t msg_len, const uint8_t sig[192], int *ok, void *stream);
int blsmi_g1pubs_verify_aggregate_common_with_domain_dev(const void *d_pks /* n*96 */, size_t n, const uint8_t msg32[32], const uint8_t domain[8], const uint8_t sig[192], int *ok, void *stream);

/* ---- Deserialize + Verify in one pass: keys and signatures in the compressed wire format (what
 * PublicKey.Serialize / Signature.Serialize produce, g2pubs/bls.go:18-20, 67-69): g2pubs pk 96 B + sig 48 B,
 * g1pubs pk 48 B + sig 96 B.  check_subgroup != 0 applies the subgroup test of DeserializePublicKey /
 * DeserializeSignature.  ok[i] = 0 when either element fails to deserialise or is the point at infinity;
 * err_pk / err_sig (n bytes each, may be NULL) receive the error codes of blsmi_g*_decompress_batch. */
int blsmi_g2pubs_verify_serialized_batch(const uint8_t *msgs, const uint64_t *off /* n+1 */, const uint8_t *pks /* n*96 */, const uint8_t *sigs /* n*48 */,
                                         int check_subgroup, uint8_t *ok /* n */, uint8_t *err_pk, uint8_t *err_sig, size_t n);
int blsmi_g1pubs_verify_serialized_batch(const uint8_t *msgs, const uint64_t *off /* n+1 */, const uint8_t *pks /* n*48 */, const uint8_t *sigs /* n*96 */,
                                         int check_subgroup, uint8_t *ok /* n */, uint8_t *err_pk, uint8_t *err_sig, size_t n);

/* ---- wire format (CompressG1/G2, DecompressG1/G2 incl. subgroup check; g1.go:185-249, g2.go:219-295)
 * err[i]: 0 ok, 1 unexpected compression mode, 2 bad infinity encoding, 3 not on curve, 4 not in subgroup  * check_subgroup: 0 = none, 1 = the subgroup test through the curve endomorphisms (the same predicate as the
 * reference's r*P == infinity for every point on the curve, 4x cheaper), 2 = the r*P form itself. */
int blsmi_g1_decompress_batch(const uint8_t *in /* n*48 */, int check_subgroup, uint8_t *out /* n*96 */, uint8_t *out_inf, uint8_t *err, size_t n);
int blsmi_g2_decompress_batch(const uint8_t *in /* n*96 */, int check_subgroup, uint8_t *out /* n*192 */, uint8_t *out_inf, uint8_t *err, size_t n);
int blsmi_g1_compress_batch(const uint8_t *pts, const uint8_t *in_inf, uint8_t *out /* n*48 */, size_t n);
int blsmi_g2_compress_batch(const uint8_t *pts, const uint8_t *in_inf, uint8_t *out /* n*96 */, size_t n);

/* ---- unit-level device ops, exported for the parity tests (tests/test_gpu_field.py).  Operands are
 * arrays of n records of `width` Fq values, each Fq as 6 LE uint64 Montgomery(2^384) limbs. -------- */
enum blsmi_debug_op {
    BLSMI_OP_FQ_MUL = 1, BLSMI_OP_FQ_SQR, BLSMI_OP_FQ_ADD, BLSMI_OP_FQ_SUB, BLSMI_OP_FQ_NEG, BLSMI_OP_FQ_INV, BLSMI_OP_FQ_SQRT,
    BLSMI_OP_FQ_DBL /* DoubleAssign fq.go:140-143 */, BLSMI_OP_FQ_CMP /* Cmp fq.go:134-137: flag = 0/1/2 for a <,=,> b */, BLSMI_OP_FQ_PARITY /* fq.go:269-273: flag */,
    BLSMI_OP_FQ2_MUL = 16, BLSMI_OP_FQ2_SQR, BLSMI_OP_FQ2_INV, BLSMI_OP_FQ2_MUL_NR, BLSMI_OP_FQ2_SQRT, BLSMI_OP_FQ2_SQRT_ANY /* either root */, BLSMI_OP_FQ2_PARITY /* fq2.go:256-260: flag */,
    BLSMI_OP_FQ6_MUL = 32, BLSMI_OP_FQ6_SQR, BLSMI_OP_FQ6_INV, BLSMI_OP_FQ6_FROB1,
    BLSMI_OP_FQ6_MUL_BY_1 /* fq6.go:40-57, c1 = b[0..1] */, BLSMI_OP_FQ6_MUL_BY_01 /* fq6.go:60-90, (c0, c1) = b[0..3] */,
    BLSMI_OP_FQ12_MUL = 48, BLSMI_OP_FQ12_SQR, BLSMI_OP_FQ12_INV, BLSMI_OP_FQ12_FROB1, BLSMI_OP_FQ12_FROB2, BLSMI_OP_FQ12_FROB3, BLSMI_OP_FQ12_CYCLO_SQR, BLSMI_OP_FQ12_CYCLO_RUN16 /* 16 squarings in compressed form + decompression */,
    BLSMI_OP_FQ12_MUL_BY_014 /* fq12.go:32-47, (c0, c1, c4) = b[0..5] */, BLSMI_OP_FQ12_MUL_BY_LINE_PAIR /* a * (014 element b[0..5]) * (014 element b[6..11]) through the fused two-line product */,
    BLSMI_OP_G1_DOUBLE = 64, BLSMI_OP_G1_ADD, BLSMI_OP_G2_DOUBLE, BLSMI_OP_G2_ADD,
    BLSMI_OP_SWU_G1 = 68 /* t in word 0 of a 3-Fq record -> (x, y, 0) */, BLSMI_OP_SWU_G2 /* t in words 0-1 of a 6-Fq record -> (x, y, 0) */,
    /* with BLSMI_OP_LANE_ROW only: one step of the homogeneous Miller loop on a 12-Fq record (X, Y, Z of the running point: Fq2 each; xq, yq of Q:
     * Fq2 each; xP, yP: Fq each) -> (X3, Y3, Z3, c0, c1, c4): the new point and the line at P.  _REF: the lane-pair routine the row form restates */
    BLSMI_OP_ROW_DBL_STEP = 80, BLSMI_OP_ROW_DBL_STEP_REF, BLSMI_OP_ROW_ADD_STEP, BLSMI_OP_ROW_ADD_STEP_REF,
    /* with BLSMI_OP_LANE_ROW only: G2 Jacobian arithmetic of the row layout's HashG2 tail (row_g2.inc) on a 12-Fq record (X1, Y1, Z1, X2, Y2, Z2: Fq2 each)
     * -> (X3, Y3, Z3, 0, 0, 0): g2.go:389-443 of the first point, g2.go:446-529 of both WITHOUT the special cases (infinity or equal x give Z3 = 0),
     * hash.go:368-389 of the first point */
    BLSMI_OP_ROW_G2_DOUBLE = 84, BLSMI_OP_ROW_G2_ADD, BLSMI_OP_ROW_CLEAR_H2
};
#define BLSMI_OP_LANE_PAIR 0x100 /* OR into an FQ2 / FQ6 / FQ12 op: run it in the lane-pair layout of the pairing kernels */
#define BLSMI_OP_LANE_QUAD 0x200 /* OR into an FQ12 op: run it in the lane-quad layout (four lanes per tuple, k_pairing_quad.hip) */
#define BLSMI_OP_LANE_ROW 0x400  /* OR into an FQ12 op (not CYCLO_RUN16): run it in the lane-row layout (sixteen lanes per tuple, k_pairing_row.hip) */
int blsmi_debug_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, uint8_t *flag /* n, may be NULL */, size_t n);
/* G2AffineToPrepared (g2.go:650-801) of one affine G2 point: 68 line-coefficient triples, each Fq2 as 12 LE uint64 Montgomery(2^384)
 * limbs, in Miller-loop order.  mode 0: computed by the one-tuple-per-lane doubling/addition steps; 1: by the lane-pair
 * steps; 2: the table of the G2 generator the library prepared at start-up for g2pubs.Verify (g2_aff ignored). */
int blsmi_debug_g2_prepare(const uint8_t *g2_aff /* 192 */, int mode, uint64_t *out /* 68*3*12 */);

/* The two halves of a small-batch hash-to-curve, for the parity tests.  kind 0 / 1 / 2 = HashG1 (hash.go:326-331) / HashG2
 * (hash.go:405-411) / HashG2WithDomain (g2.go:1041-1085).
 * blsmi_debug_hash_tail: the curve arithmetic after the SWU maps (isogeny, sum, cofactor clearing; kind 2: ScaleByCofactor)
 * run as a level program of the latency path.  pts: per message the mapped affine points as big-endian 48-byte field
 * elements -- kind 0: two points of the 11-isogenous curve (192 bytes), kind 1: two points of the 3-isogenous curve (384),
 * kind 2: the point of E'(Fq2) the try-and-increment search found (192).  out: the hash point (96 / 192 / 192 bytes);
 * good[i] = 0 when message i met an exceptional step (the library then re-hashes it with the kernel below).
 * blsmi_debug_hash_redo: that kernel -- messages with good[i] == 0 are hashed by the one-message-per-lane routine into
 * out[i]; the records of the others are left as they were.  n <= 4096 for the tail. */
int blsmi_debug_hash_tail(int kind, const uint8_t *pts, uint8_t *out, uint8_t *good, size_t n);
/* the same tail of HashG1 as the THROUGHPUT kernel runs it (k_hash_g1_finish), with (clear != 0) or without the cofactor clearing of hash.go:306-309;
 * *special gets bit 1 when clear == 0 and some message's two mapped points cancel -- the case a large VerifyAggregate's uncleared-hash path hands
 * back to the cleared one. */
int blsmi_debug_hash_g1_finish(const uint8_t *pts /* n*192 */, int clear /* 2: the four-lanes-per-message tail + its redo pass; *special = messages redone */, uint8_t *out /* n*96 */, int *special, size_t n);
int blsmi_debug_hash_redo(int kind, const uint8_t *msgs, const uint64_t *off_or_domain, const uint8_t *good, uint8_t *out /* in/out */, size_t n);

#ifdef __cplusplus
}
#endif
#endif
