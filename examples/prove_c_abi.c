/* prove_c_abi.c -- the drop-in boundary used from plain C (no Python, no torch).
 *
 * Mirrors main.rs:15-36 of the reference: prove, verify, print the proof size.
 *   gcc -O2 -Iinclude examples/prove_c_abi.c -Lzkstark_amd -lzkstark_amd -Wl,-rpath,$PWD/zkstark_amd -o prove_c_abi
 *   ./prove_c_abi [log_n] [log_blowup]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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
    uint32_t log_n = argc > 1 ? (uint32_t)atoi(argv[1]) : 10;      /* prover.rs:32, :48: trace 1023, group 1024 */
    uint32_t log_b = argc > 2 ? (uint32_t)atoi(argv[2]) : 3;       /* prover.rs:49: domain 8192 */
    if (zk_abi_version() != ZK_ABI_VERSION) {                           /* the library on the path was built from another zkstark_amd.h */
        fprintf(stderr, "libzkstark_amd speaks ABI version %u, this program was compiled against %u\n", zk_abi_version(), ZK_ABI_VERSION);
        return 2;
    }
    size_t n = (size_t)1 << log_n;
    uint32_t *trace = malloc((n - 1) * sizeof *trace);
    CHECK(zk_trace_fibsq(1, 3141592, n - 1, trace));               /* prover.rs:32-39 */
    printf("a[n-2] = %u\n", trace[n - 2]);                          /* prover.rs:42: 2338775057 at n = 1024 */

    zk_ctx *ctx = NULL;
    CHECK(zk_ctx_create(0, log_n, log_b, &ctx));
    size_t cap = zk_proof_data_len(log_n, log_b), len = 0;
    uint8_t *proof = malloc(cap), state[32];
    double t0 = now_ms();
    CHECK(zk_prove(ctx, trace, n - 1, proof, cap, &len, state));   /* main.rs:23: generate_proof(channel) */
    printf("Prover runtime: %.3f ms (context setup %.1f ms, not included)\n", now_ms() - t0, zk_ctx_setup_ms(ctx));
    t0 = now_ms();
    CHECK(zk_verify(proof, len, log_n, log_b, trace[n - 2]));      /* main.rs:28: proof.verify() */
    CHECK(zk_verify_strict(proof, len, state, log_n, log_b, trace[n - 2]));
    printf("Verifier runtime: %.3f ms\n", now_ms() - t0);
    printf("Proof size: %zu\n", zk_proof_size(len));                /* main.rs:35 */
    printf("proof head:");
    for (int i = 0; i < 8; ++i) printf(" %02x", proof[i]);
    printf("\nfinal state:");
    for (int i = 0; i < 8; ++i) printf(" %02x", state[i]);
    printf("\n");
    zk_ctx_destroy(ctx);
    free(proof);
    free(trace);
    return 0;
}
