/* A plain-C99 caller of the C ABI -- what cgo compiles from the shim of INTEGRATION.md, minus Go: no Python, no C++, only
 * include/blsmi.h and -lblsmi.  Reads tuples written by tests/test_gpu_round2.py (count, then per tuple: message length,
 * message, 192-byte public key, 96-byte signature), calls g2pubs Verify one tuple per call (g2pubs/bls.go:159-162) and then
 * as one batch, and one Pairing; prints the verdicts and the pairing's 576 bytes in hex for the test to compare.
 *   gcc -std=c99 -O2 -I include tests/native/abi_client.c -o <out> -L bls_amd -lblsmi -Wl,-rpath,<abs bls_amd> */
#include "blsmi.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
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
