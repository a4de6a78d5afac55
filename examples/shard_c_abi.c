/* shard_c_abi.c -- the sharded prover from plain C (no Python, no torch): one thread per GPU, RCCL inside the library.
 *
 * generate_proof (prover.rs:9-293) for one proof whose evaluation domain is distributed over `world` GPUs of this node
 * (include/zkstark_amd.h: zk_shard_*).  Rank 0 obtains the RCCL unique id, every thread creates its shard collectively,
 * uploads the same trace and proves; every rank must return the same bytes, equal to zk_prove's on one GPU.
 *   gcc -O2 -pthread -Iinclude examples/shard_c_abi.c -Lzkstark_amd -lzkstark_amd -Wl,-rpath,$PWD/zkstark_amd -o shard_c_abi
 *   ./shard_c_abi [world] [log_n] [log_blowup] [peer]  (world GPUs must be visible; world = 1 exercises RCCL with one rank;
 *                                                        a 4th argument "peer" uses the library's peer-copy transport instead: no RCCL)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkstark_amd.h"

static int g_peer_copy;       /* argv[4] == "peer": zk_shard_options.peer_copy (IPC handles + device-to-device copies, csrc/peer.hpp) */

typedef struct {
    int rank, world, rc;
    uint32_t log_n, log_b;
    const uint8_t *id;
    const uint32_t *trace;
    uint8_t *proof, state[32];
    size_t cap, len;
    zk_shard_stats stats;
    char err[256];
} rank_args;

static void *run_rank(void *p) {
    rank_args *a = p;
    zk_shard *sp = NULL;
    zk_shard_options opt;
    ZK_STRUCT_INIT(&opt);                               /* zeroes it and sets opt.struct_size = sizeof opt (checked by the library) */
    ZK_STRUCT_INIT(&a->stats);
    opt.force_collectives = 1;                          /* with world = 1: still go through the transport */
    opt.peer_copy = g_peer_copy;
    if (a->log_n + a->log_b < 22) { opt.min_layer_log = 1; opt.min_chunk_log = 6; }   /* small demo sizes: shard anyway */
    /* one GPU per rank; ZK_EXAMPLE_SHARE_GPU=1 puts every rank on GPU 0 (a one-GPU box: only the peer-copy transport can do that) */
    const int gpu = getenv("ZK_EXAMPLE_SHARE_GPU") ? 0 : a->rank;
    a->rc = zk_shard_create(gpu, a->rank, a->world, a->id, NULL /* built-in transport: RCCL, or peer copies */, &opt, a->log_n, a->log_b, &sp);
    if (!a->rc) a->rc = zk_shard_trace_upload(sp, a->trace, ((size_t)1 << a->log_n) - 1);
    if (!a->rc) a->rc = zk_shard_prove(sp, a->proof, a->cap, &a->len, a->state);   /* collective */
    if (!a->rc) a->rc = zk_shard_get_stats(sp, &a->stats);
    if (a->rc) snprintf(a->err, sizeof a->err, "%s", zk_last_error());
    zk_shard_destroy(sp);
    return NULL;
}

int main(int argc, char **argv) {
    int world = argc > 1 ? atoi(argv[1]) : 1;
    uint32_t log_n = argc > 2 ? (uint32_t)atoi(argv[2]) : 12, log_b = argc > 3 ? (uint32_t)atoi(argv[3]) : 3;
    if (world < 1 || world > 8) { fprintf(stderr, "world must be 1..8\n"); return 2; }
    if (zk_abi_version() != ZK_ABI_VERSION) {           /* the library on the path was built from another zkstark_amd.h */
        fprintf(stderr, "libzkstark_amd speaks ABI version %u, this program was compiled against %u\n", zk_abi_version(), ZK_ABI_VERSION);
        return 2;
    }
    size_t n = (size_t)1 << log_n, cap = zk_proof_data_len(log_n, log_b);
    uint32_t *trace = malloc((n - 1) * sizeof *trace);
    if (zk_trace_fibsq(1, 3141592, n - 1, trace)) return 1;          /* prover.rs:32-39 */
    uint8_t id[ZK_SHARD_ID_BYTES];
    g_peer_copy = argc > 4 && !strcmp(argv[4], "peer");
    if (g_peer_copy) { for (int i = 0; i < ZK_SHARD_ID_BYTES; ++i) id[i] = (uint8_t)(rand() ^ (i * 41)); }   /* any shared bytes: they name the page */
    else if (zk_shard_unique_id(id)) { fprintf(stderr, "zk_shard_unique_id: %s\n", zk_last_error()); return 1; }
    rank_args args[8];
    pthread_t th[8];
    for (int r = 0; r < world; ++r) {
        memset(&args[r], 0, sizeof args[r]);
        args[r].rank = r; args[r].world = world; args[r].log_n = log_n; args[r].log_b = log_b;
        args[r].id = id; args[r].trace = trace; args[r].cap = cap; args[r].proof = malloc(cap);
        pthread_create(&th[r], NULL, run_rank, &args[r]);
    }
    for (int r = 0; r < world; ++r) pthread_join(th[r], NULL);
    for (int r = 0; r < world; ++r)
        if (args[r].rc) { fprintf(stderr, "rank %d: %d: %s\n", r, args[r].rc, args[r].err); return 1; }
    for (int r = 1; r < world; ++r)
        if (args[r].len != args[0].len || memcmp(args[r].proof, args[0].proof, args[0].len) || memcmp(args[r].state, args[0].state, 32)) {
            fprintf(stderr, "rank %d disagrees with rank 0\n", r);
            return 1;
        }
    /* the single-GPU prover on the same trace: the sharded proof must be byte-identical */
    zk_ctx *ctx = NULL;
    uint8_t *one = malloc(cap), st1[32];
    size_t len1 = 0;
    if (zk_ctx_create(0, log_n, log_b, &ctx) || zk_prove(ctx, trace, n - 1, one, cap, &len1, st1)) { fprintf(stderr, "%s\n", zk_last_error()); return 1; }
    zk_ctx_destroy(ctx);
    if (len1 != args[0].len || memcmp(one, args[0].proof, len1) || memcmp(st1, args[0].state, 32)) { fprintf(stderr, "sharded proof differs from zk_prove\n"); return 1; }
    if (zk_verify_strict(args[0].proof, args[0].len, args[0].state, log_n, log_b, trace[n - 2])) { fprintf(stderr, "%s\n", zk_last_error()); return 1; }
    printf("world %d: %zu proof bytes on every rank, equal to zk_prove; verifier accepts\n", world, args[0].len);
    printf("sharded layers %u, native rccl %u, peer copy %u, root board %u, sent to peers %.0f bytes per rank\n", args[0].stats.sharded_layers,
           args[0].stats.native_rccl, args[0].stats.peer_copy, args[0].stats.root_board, args[0].stats.sent_bytes);
    printf("proof head:");
    for (int i = 0; i < 8; ++i) printf(" %02x", args[0].proof[i]);
    printf("\n");
    return 0;
}
