/* batch_c_abi.c -- many proofs of the reference's size in one go, from plain C.
 *
 * The reference proves one trace per run (main.rs:15-36); this proves 2^log_batch of them in lockstep
 * (zk_batch_*), verifies every proof with the reference's checks (proof.rs:15) and prints the rate.
 *   gcc -O2 -Iinclude examples/batch_c_abi.c -Lzkstark_amd -lzkstark_amd -Wl,-rpath,$PWD/zkstark_amd -o batch_c_abi
 *   ./batch_c_abi [log_batch]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zkstark_amd.h"

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != ZK_OK) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, zk_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char **argv) {
    const uint32_t log_n = 10, log_b = 3;                              /* prover.rs:48-57 */
    uint32_t log_batch = argc > 1 ? (uint32_t)atoi(argv[1]) : 6;
    if (zk_abi_version() != ZK_ABI_VERSION) {                           /* the library on the path was built from another zkstark_amd.h */
        fprintf(stderr, "libzkstark_amd speaks ABI version %u, this program was compiled against %u\n", zk_abi_version(), ZK_ABI_VERSION);
        return 2;
    }
    zk_batch *b = NULL;
    CHECK(zk_batch_create(0, log_n, log_b, log_batch, &b));
    size_t batch = zk_batch_size(b), plen = zk_proof_data_len(log_n, log_b);
    uint32_t *a0 = malloc(batch * 4), *a1 = malloc(batch * 4), *last = malloc(batch * 4);
    for (size_t p = 0; p < batch; ++p) { a0[p] = 1; a1[p] = 3141592 + (uint32_t)p; }   /* proof 0 is the reference's trace */
    CHECK(zk_batch_gen_fibsq(b, a0, a1));                              /* prover.rs:32-39 for every proof, on the device */
    CHECK(zk_batch_public_last(b, last));
    printf("a[n-2] of proof 0 = %u\n", last[0]);                        /* prover.rs:42 */
    uint8_t *proofs = malloc(batch * plen), *states = malloc(batch * 32);
    CHECK(zk_batch_prove(b, proofs, plen, states));                    /* warm-up */
    double t0 = now_ms();
    CHECK(zk_batch_prove(b, proofs, plen, states));
    double ms = now_ms() - t0;
    printf("%zu proofs in %.3f ms: %.2f us per proof\n", batch, ms, ms * 1e3 / (double)batch);
    for (size_t p = 0; p < batch; ++p) {
        CHECK(zk_verify(proofs + p * plen, plen, log_n, log_b, last[p]));               /* main.rs:28 */
        CHECK(zk_verify_strict(proofs + p * plen, plen, states + 32 * p, log_n, log_b, last[p]));
    }
    printf("all %zu proofs verified; proof size %zu\n", batch, zk_proof_size(plen));   /* main.rs:35 */
    printf("proof 0 head:");
    for (int i = 0; i < 8; ++i) printf(" %02x", proofs[i]);
    printf("\n");
    zk_batch_destroy(b);
    free(a0); free(a1); free(last); free(proofs); free(states);
    return 0;
}
