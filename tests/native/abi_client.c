/* A plain-C99 caller of the C ABI -- what cgo compiles from the shim of INTEGRATION.md, minus Go: no Python, no C++, only
 * include/blsmi.h and -lblsmi.  Reads tuples written by tests/test_gpu_round2.py (count, then per tuple: message length,
 * message, 192-byte public key, 96-byte signature), calls g2pubs Verify one tuple per call (g2pubs/bls.go:159-162) and then
 * as one batch, and one Pairing; prints the verdicts and the pairing's 576 bytes in hex for the test to compare.
 *   gcc -std=c99 -O2 -I include tests/native/abi_client.c -o <out> -L bls_amd -lblsmi -Wl,-rpath,<abs bls_amd>
 * `abi_client jac <file>` is the leg the Go shims actually use since blsmi 0.6: the points are handed over as C
 * structs with the layout of *bls.G1Projective / *bls.G2Projective (g1.go:252-256, g2.go:298-302: x, y, z; FQ2 = two FQ; FQ = 6 uint64,
 * Montgomery 2^384, little-endian limbs) -- 18 / 36 words -- exactly what `(*C.uint64_t)(unsafe.Pointer(p))` passes from Go. */
#include "blsmi.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the Go structs, restated in C (no padding: every member is an array of uint64_t) */
struct fq { uint64_t l[6]; };
struct fq2 { struct fq c0, c1; };
struct g1_projective { struct fq x, y, z; };
struct g2_projective { struct fq2 x, y, z; };
typedef char g1_projective_is_18_words[sizeof(struct g1_projective) == 144 ? 1 : -1];
typedef char g2_projective_is_36_words[sizeof(struct g2_projective) == 288 ? 1 : -1];

static void print_words(const char *tag, const uint64_t *w, size_t n) {
    printf("%s", tag);
    for (size_t i = 0; i < n; i++) printf(" %016llx", (unsigned long long)w[i]);
    printf("\n");
}

/* file: n, m (the first m tuples are untouched: their signatures aggregate to a valid VerifyAggregate), then per tuple message length,
 * message, G2Projective public key, G1Projective signature, 32-byte secret key (big-endian) */
static int jac_leg(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror("open"); return 2; }
    uint64_t n = 0, m = 0;
    if (fread(&n, 8, 1, f) != 1 || fread(&m, 8, 1, f) != 1 || n == 0 || n > 4096 || m > n) return 2;
    uint8_t *msgs = malloc(1 << 20), *sks = malloc(32 * n), *ok = malloc(n);
    uint64_t *off = malloc(8 * (n + 1));
    struct g2_projective *pks = malloc(sizeof(struct g2_projective) * n);
    struct g1_projective *sigs = malloc(sizeof(struct g1_projective) * n), *made = malloc(sizeof(struct g1_projective) * n);
    off[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t len = 0;
        if (fread(&len, 4, 1, f) != 1 || off[i] + len > (1u << 20)) return 2;
        if (len && fread(msgs + off[i], 1, len, f) != len) return 2;
        off[i + 1] = off[i] + len;
        if (fread(&pks[i], sizeof pks[i], 1, f) != 1 || fread(&sigs[i], sizeof sigs[i], 1, f) != 1 || fread(sks + 32 * i, 1, 32, f) != 32) return 2;
    }
    fclose(f);
    int rc = blsmi_init(0);
    if (rc != BLSMI_OK) { fprintf(stderr, "blsmi_init: %d\n", rc); return 1; }
    /* VerifyBatch as the g2pubs shim calls it: packKeys / packSigs are one struct copy per point */
    rc = blsmi_g2pubs_verify_batch_jac(msgs, off, (const uint64_t *)pks, (const uint64_t *)sigs, ok, NULL, (size_t)n);
    if (rc != BLSMI_OK) { fprintf(stderr, "verify_batch_jac: %d\n", rc); return 1; }
    printf("jacbatch");
    for (uint64_t i = 0; i < n; i++) printf(" %d", ok[i]);
    printf("\n");
    /* SumPublicKeys / SumSignatures: the sum comes back as a struct of the same layout (z = 1) */
    struct g2_projective pksum; struct g1_projective sgsum_m, sgsum_n;
    int inf = -1;
    if ((rc = blsmi_g2_sum_jac((const uint64_t *)pks, (size_t)n, (uint64_t *)&pksum, &inf)) != BLSMI_OK) { fprintf(stderr, "g2_sum_jac: %d\n", rc); return 1; }
    printf("sumg2inf %d\n", inf);
    print_words("sumg2", (const uint64_t *)&pksum, 36);
    if ((rc = blsmi_g1_sum_jac((const uint64_t *)sigs, (size_t)m, (uint64_t *)&sgsum_m, &inf)) != BLSMI_OK) { fprintf(stderr, "g1_sum_jac: %d\n", rc); return 1; }
    if ((rc = blsmi_g1_sum_jac((const uint64_t *)sigs, (size_t)n, (uint64_t *)&sgsum_n, &inf)) != BLSMI_OK) { fprintf(stderr, "g1_sum_jac: %d\n", rc); return 1; }
    /* (*Signature).VerifyAggregate: the aggregate of the m untouched tuples verifies, the aggregate of all n (corrupted ones included) does not */
    int agg = -1;
    if ((rc = blsmi_g2pubs_verify_aggregate_jac(msgs, off, (const uint64_t *)pks, (const uint64_t *)&sgsum_m, (size_t)m, &agg)) != BLSMI_OK) { fprintf(stderr, "verify_aggregate_jac: %d\n", rc); return 1; }
    printf("aggregate_m %d\n", agg);
    if ((rc = blsmi_g2pubs_verify_aggregate_jac(msgs, off, (const uint64_t *)pks, (const uint64_t *)&sgsum_n, (size_t)n, &agg)) != BLSMI_OK) { fprintf(stderr, "verify_aggregate_jac: %d\n", rc); return 1; }
    printf("aggregate_n %d\n", agg);
    /* SignBatch: signatures handed back as G1Projective structs */
    if ((rc = blsmi_g2pubs_sign_batch_jac(msgs, off, sks, (uint64_t *)made, (size_t)n)) != BLSMI_OK) { fprintf(stderr, "sign_batch_jac: %d\n", rc); return 1; }
    print_words("signed", (const uint64_t *)made, 18 * (size_t)n);
    /* and what was just signed verifies against keys derived on the device (PrivToPub: k * G2 generator, as structs) */
    struct g2_projective *derived = malloc(sizeof(struct g2_projective) * n);
    if ((rc = blsmi_g2_mul_generator_batch_jac(sks, (uint64_t *)derived, (size_t)n)) != BLSMI_OK) { fprintf(stderr, "mul_generator_batch_jac: %d\n", rc); return 1; }
    if ((rc = blsmi_g2pubs_verify_batch_jac(msgs, off, (const uint64_t *)derived, (const uint64_t *)made, ok, NULL, (size_t)n)) != BLSMI_OK) { fprintf(stderr, "verify_batch_jac: %d\n", rc); return 1; }
    printf("roundtrip");
    for (uint64_t i = 0; i < n; i++) printf(" %d", ok[i]);
    printf("\n");
    blsmi_shutdown();
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 3 && strcmp(argv[1], "jac") == 0) return jac_leg(argv[2]);
    if (argc < 2) { fprintf(stderr, "usage: %s tuples.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1 || n == 0 || n > 4096) return 2;
    uint8_t *msgs = malloc(1 << 20), *pks = malloc(192 * n), *sigs = malloc(96 * n), *ok = malloc(n), *ok1 = malloc(n);
    uint64_t *off = malloc(8 * (n + 1));
    off[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t len = 0;
        if (fread(&len, 4, 1, f) != 1 || off[i] + len > (1u << 20)) return 2;
        if (len && fread(msgs + off[i], 1, len, f) != len) return 2;
        off[i + 1] = off[i] + len;
        if (fread(pks + 192 * i, 1, 192, f) != 192 || fread(sigs + 96 * i, 1, 96, f) != 96) return 2;
    }
    fclose(f);
    int rc = blsmi_init(0);
    if (rc != BLSMI_OK) { fprintf(stderr, "blsmi_init: %d\n", rc); return 1; }
    printf("version %s devices %d\n", blsmi_version(), blsmi_device_count());
    for (uint64_t i = 0; i < n; i++) {                       /* the Go API's shape: one tuple per call */
        uint64_t o2[2] = {0, off[i + 1] - off[i]};
        rc = blsmi_g2pubs_verify_batch(msgs + off[i], o2, pks + 192 * i, sigs + 96 * i, NULL, ok1 + i, NULL, 1);
        if (rc != BLSMI_OK) { fprintf(stderr, "verify: %d\n", rc); return 1; }
    }
    rc = blsmi_g2pubs_verify_batch(msgs, off, pks, sigs, NULL, ok, NULL, (size_t)n);
    if (rc != BLSMI_OK) { fprintf(stderr, "verify batch: %d\n", rc); return 1; }
    printf("single");
    for (uint64_t i = 0; i < n; i++) printf(" %d", ok1[i]);
    printf("\nbatch");
    for (uint64_t i = 0; i < n; i++) printf(" %d", ok[i]);
    printf("\n");
    /* the same tuples over PREPARED keys (blsmi 0.4): the tables live in device memory the library owns; tuple i uses table i */
    void *prepared = NULL;
    uint8_t *okp = malloc(n);
    rc = blsmi_g2_prepared_create(pks, (size_t)n, &prepared);
    if (rc != BLSMI_OK || !prepared) { fprintf(stderr, "prepared_create: %d\n", rc); return 1; }
    rc = blsmi_g2pubs_verify_batch_prepared(msgs, off, prepared, NULL, sigs, NULL, okp, NULL, (size_t)n);
    if (rc != BLSMI_OK) { fprintf(stderr, "verify prepared: %d\n", rc); return 1; }
    printf("prepared");
    for (uint64_t i = 0; i < n; i++) printf(" %d", okp[i]);
    printf("\n");
    if ((rc = blsmi_g2_prepared_destroy(prepared)) != BLSMI_OK) { fprintf(stderr, "prepared_destroy: %d\n", rc); return 1; }
    /* and out of page-locked staging buffers (blsmi_host_alloc): what a shim that serialises straight into them hands over */
    void *spk = NULL, *ssg = NULL, *sok = NULL;
    if (blsmi_host_alloc(192 * n, &spk) != BLSMI_OK || blsmi_host_alloc(96 * n, &ssg) != BLSMI_OK || blsmi_host_alloc(n, &sok) != BLSMI_OK || !spk || !ssg || !sok) { fprintf(stderr, "host_alloc\n"); return 1; }
    memcpy(spk, pks, 192 * n); memcpy(ssg, sigs, 96 * n);
    rc = blsmi_g2pubs_verify_batch(msgs, off, (const uint8_t *)spk, (const uint8_t *)ssg, NULL, (uint8_t *)sok, NULL, (size_t)n);
    if (rc != BLSMI_OK) { fprintf(stderr, "verify from page-locked buffers: %d\n", rc); return 1; }
    printf("pinned");
    for (uint64_t i = 0; i < n; i++) printf(" %d", ((uint8_t *)sok)[i]);
    printf("\n");
    if (blsmi_host_free(spk) != BLSMI_OK || blsmi_host_free(ssg) != BLSMI_OK || blsmi_host_free(sok) != BLSMI_OK || blsmi_host_free(NULL) != BLSMI_OK) { fprintf(stderr, "host_free\n"); return 1; }
    uint64_t e[72];
    rc = blsmi_pairing_batch(sigs, pks, e, 1);               /* e(sig_0, pk_0) */
    if (rc != BLSMI_OK) { fprintf(stderr, "pairing: %d\n", rc); return 1; }
    printf("pairing");
    for (int i = 0; i < 72; i++) printf(" %016llx", (unsigned long long)e[i]);
    printf("\n");
    blsmi_shutdown();
    return 0;
}
