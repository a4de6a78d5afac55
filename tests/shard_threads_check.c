/* shard_threads_check.c -- the native sharded prover with `world` ranks as THREADS of one process on one GPU
 * (tests/test_gpu_shard_native.py).  A one-GPU box allows only a handful of processes on the card, so this is how
 * world = 8 (lg = 3: the full top path, local blow-up 1, eight-way all-to-all) is exercised natively; it is also the
 * threading model of examples/shard_c_abi.c.  The transport is the caller's (include/zkstark_amd.h:
 * zk_shard_transport): peers publish their send pointers, a barrier, device-to-device copies, a barrier.
 *   gcc -O2 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/shard_threads_check.c \
 *       -Lzkstark_amd -lzkstark_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zkstark_amd -Wl,-rpath,/opt/rocm/lib -o shard_threads_check
 *   ./shard_threads_check world log_n log_blowup min_layer_log min_chunk_log overlap_min_log [timed_reps [fail_rank]]
 * A harness binary does not outlive the library it was compiled against: it refuses to run when the library's ABI version
 * differs from its header's, and -- when compiled with -DZK_EXPECT_BUILD_HASH=\"<zk_build_hash of the library>\" (the
 * tests and tools/run_shard_threads.sh do) -- when the library was built from other sources.  Every caller-allocated
 * struct carries its size (ZK_STRUCT_INIT), so a stale binary gets ZK_ERR_INVALID instead of shifted fields.  Every
 * host-side wait of the library is bounded by opt.timeout_s = 20 s (environment ZK_HARNESS_TIMEOUT_S), so that a stuck
 * exchange names rank, peer and layer on stderr well before an outer `timeout` would kill the process silently.
 */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zkstark_amd.h"

#define MAXW 32
static int g_world;
/* A barrier the ranks can be released from: a rank that fails sets g_abort and never arrives, the others leave the
 * barrier with an error instead of waiting for it for ever (the transport then reports the failure to the library). */
static atomic_int g_abort, g_count, g_gen;
static int g_fail_rank = -1;                  /* argv[8]: this rank injects a failure instead of proving */
static double g_timeout_s = 20.0;             /* also zk_shard_options.timeout_s of every rank */
static double mono_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static int bar_wait(void) {
    const int gen = atomic_load(&g_gen);
    if (atomic_fetch_add(&g_count, 1) + 1 == g_world) { atomic_store(&g_count, 0); atomic_fetch_add(&g_gen, 1); return 0; }
    const double t0 = mono_s();
    unsigned spins = 0;
    while (atomic_load(&g_gen) == gen) {
        if (atomic_load(&g_abort)) return 1;
        if ((++spins & 1023) == 0 && mono_s() - t0 > g_timeout_s) {   /* the harness's own waits are bounded like the library's */
            fprintf(stderr, "harness barrier #%d: %d of %d ranks arrived within %.0f s; giving up\n", gen, atomic_load(&g_count), g_world, g_timeout_s);
            atomic_store(&g_abort, 1);
            return 1;
        }
        sched_yield();
    }
    return 0;
}
static const uint32_t *const *g_send[MAXW];   /* rank r's send pointer table for the exchange in progress */
static const uint32_t *g_gather[MAXW];

typedef struct { int rank; unsigned calls; } tp_user;

/* ---- stream-ordered variant of the transport (environment ZK_HARNESS_ASYNC=1; timing runs) ------------------------------
 * The default transport above synchronises the device twice per exchange, so the host threads meet at four points per
 * collective and the summed device time it reports moves with the host (45 .. 51 ms for the same eight-rank proof from session to
 * session).  This one behaves like RCCL does: nothing waits on the host.  A rank records "my pieces are ready" on the stream it
 * was given, the ranks meet at a host barrier only to exchange pointers and event handles, every rank's stream then waits for
 * its peers' events, copies its pieces, records "I have read", and waits for the peers' "have read" events before it goes on
 * (so that no send buffer is overwritten before it has been read).  Events come from a ring per rank, indexed by the number
 * of the collective (every rank issues the same collectives in the same order). */
#define EV_RING 256
static int g_async;
static hipEvent_t g_ready[MAXW][EV_RING], g_done[MAXW][EV_RING];
static int async_events(int me) {
    for (int k = 0; k < EV_RING; ++k)
        if (hipEventCreateWithFlags(&g_ready[me][k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g_done[me][k], hipEventDisableTiming) != hipSuccess) return 1;
    return 0;
}
static int async_exchange(tp_user *u, hipStream_t st, int gather, const uint32_t *const *send, const uint32_t *gsend, uint32_t *const *recv,
                          uint32_t *grecv, size_t words) {
    const int me = u->rank, k = (int)(u->calls++ % EV_RING);
    if (hipEventRecord(g_ready[me][k], st) != hipSuccess) return 1;
    if (gather) g_gather[me] = gsend; else g_send[me] = send;
    if (bar_wait()) return 2;                     /* pointers published, "ready" events recorded: no device synchronisation */
    int bad = 0;
    for (int q = 0; q < g_world; ++q) {
        if (q != me && hipStreamWaitEvent(st, g_ready[q][k], 0) != hipSuccess) bad = 1;
        const void *src = gather ? (const void *)g_gather[q] : (const void *)g_send[q][me];
        void *dst = gather ? (void *)(grecv + (size_t)q * words) : (void *)recv[q];
        if (hipMemcpyAsync(dst, src, words * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) bad = 1;
    }
    if (hipEventRecord(g_done[me][k], st) != hipSuccess) bad = 1;
    if (bar_wait()) return 2;                     /* every rank's "have read" event is recorded */
    for (int q = 0; q < g_world; ++q)
        if (q != me && hipStreamWaitEvent(st, g_done[q][k], 0) != hipSuccess) bad = 1;
    return bad;
}

static int tp_all_to_all(void *user, const uint32_t *const *send, uint32_t *const *recv, size_t words, void *stream) {
    const int me = ((tp_user *)user)->rank;
    if (g_async) return async_exchange((tp_user *)user, (hipStream_t)stream, 0, send, NULL, recv, NULL, words);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;   /* my send pieces are complete */
    g_send[me] = send;
    if (bar_wait()) return 2;
    int bad = 0;
    for (int q = 0; q < g_world; ++q)   /* piece `me` of rank q's table comes to my recv[q] */
        if (hipMemcpyAsync(recv[q], g_send[q][me], words * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) bad = 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) bad = 1;
    if (bar_wait()) return 2;           /* nobody reuses a send buffer before every peer has read it */
    return bad;
}
static int tp_all_gather(void *user, const uint32_t *send, uint32_t *recv, size_t words, void *stream) {
    const int me = ((tp_user *)user)->rank;
    if (g_async) return async_exchange((tp_user *)user, (hipStream_t)stream, 1, NULL, send, NULL, recv, words);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    g_gather[me] = send;
    if (bar_wait()) return 2;
    int bad = 0;
    for (int q = 0; q < g_world; ++q)
        if (hipMemcpyAsync(recv + (size_t)q * words, g_gather[q], words * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) bad = 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) bad = 1;
    if (bar_wait()) return 2;
    return bad;
}

typedef struct {
    int rank, rc;
    uint32_t log_n, log_b;
    zk_shard_options opt;
    const uint8_t *id;
    const uint32_t *trace;
    uint8_t *proof, state[32], root[32];
    size_t cap, len;
    zk_shard_stats stats;
    int reps;               /* extra timed proofs (argv[7]); the time shows the TOTAL device work of all ranks on one GPU */
    double ms_per_proof;
    char err[256];
} rank_args;

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static void *run_rank(void *p) {
    rank_args *a = p;
    tp_user u = {a->rank, 0};
    if (g_async && (hipSetDevice(0) != hipSuccess || async_events(a->rank))) { a->rc = ZK_ERR_HIP; snprintf(a->err, sizeof a->err, "event creation failed"); atomic_store(&g_abort, 1); return NULL; }
    zk_shard_transport tp = {&u, tp_all_to_all, tp_all_gather};
    zk_shard *sp = NULL;
    a->stats.struct_size = (uint32_t)sizeof a->stats;
    a->rc = zk_shard_create(0, a->rank, g_world, a->id, &tp, &a->opt, a->log_n, a->log_b, &sp);
    if (!a->rc) a->rc = zk_shard_trace_upload(sp, a->trace, ((size_t)1 << a->log_n) - 1);
    if (!a->rc && a->rank == g_fail_rank) a->rc = zk_shard_inject_failure(sp, ZK_ERR_STATE);   /* leaves the proof; peers must not hang */
    if (!a->rc) a->rc = zk_shard_prove(sp, a->proof, a->cap, &a->len, a->state);
    if (!a->rc) a->rc = zk_shard_get_stats(sp, &a->stats);
    if (!a->rc && a->reps > 0) {
        if (bar_wait()) a->rc = ZK_ERR_STATE;
        const double t0 = now_ms();
        for (int i = 0; i < a->reps && !a->rc; ++i) a->rc = zk_shard_prove(sp, a->proof, a->cap, &a->len, a->state);
        if (!a->rc && bar_wait()) a->rc = ZK_ERR_STATE;
        a->ms_per_proof = (now_ms() - t0) / a->reps;
    }
    if (!a->rc) a->rc = zk_shard_lde_commit(sp, a->root);
    if (a->rc) { snprintf(a->err, sizeof a->err, "%s", zk_last_error()); atomic_store(&g_abort, 1); }   /* release the peers */
    zk_shard_destroy(sp);
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 7) { fprintf(stderr, "usage: world log_n log_blowup min_layer_log min_chunk_log overlap_min_log\n"); return 2; }
    if (zk_abi_version() != ZK_ABI_VERSION) {
        fprintf(stderr, "stale harness: libzkstark_amd speaks ABI version %u, this binary was compiled against %u -- recompile it\n", zk_abi_version(), ZK_ABI_VERSION);
        return 2;
    }
#ifdef ZK_EXPECT_BUILD_HASH
    if (strcmp(zk_build_hash(), ZK_EXPECT_BUILD_HASH)) {
        fprintf(stderr, "stale harness: compiled for library build %s, the library on the path is %s -- recompile it\n", ZK_EXPECT_BUILD_HASH, zk_build_hash());
        return 2;
    }
#endif
    if (getenv("ZK_HARNESS_TIMEOUT_S") && atof(getenv("ZK_HARNESS_TIMEOUT_S")) > 0) g_timeout_s = atof(getenv("ZK_HARNESS_TIMEOUT_S"));
    const double timeout_s = g_timeout_s;
    g_world = atoi(argv[1]);
    const uint32_t log_n = (uint32_t)atoi(argv[2]), log_b = (uint32_t)atoi(argv[3]);
    if (g_world < 1 || g_world > MAXW) return 2;
    if (argc > 8) g_fail_rank = atoi(argv[8]);
    g_async = getenv("ZK_HARNESS_ASYNC") != NULL && atoi(getenv("ZK_HARNESS_ASYNC")) != 0;
    const size_t n = (size_t)1 << log_n, cap = zk_proof_data_len(log_n, log_b);
    uint32_t *trace = malloc((n - 1) * sizeof *trace);
    if (zk_trace_fibsq(1, 3141592, n - 1, trace)) return 1;
    uint8_t id[ZK_SHARD_ID_BYTES];
    for (int i = 0; i < ZK_SHARD_ID_BYTES; ++i) id[i] = (uint8_t)(rand() ^ (i * 37));   /* names the root board */
    rank_args args[MAXW];
    pthread_t th[MAXW];
    for (int r = 0; r < g_world; ++r) {
        memset(&args[r], 0, sizeof args[r]);
        args[r].rank = r; args[r].log_n = log_n; args[r].log_b = log_b; args[r].id = id; args[r].trace = trace;
        ZK_STRUCT_INIT(&args[r].opt);
        args[r].opt.timeout_s = timeout_s;
        args[r].opt.min_layer_log = (uint32_t)atoi(argv[4]); args[r].opt.min_chunk_log = (uint32_t)atoi(argv[5]);
        args[r].opt.overlap_min_log = (uint32_t)atoi(argv[6]);
        args[r].opt.exchange_cp = getenv("ZK_HARNESS_EXCHANGE_CP") ? 1 : 0;   /* A/B: cp exchanged like every other layer (rounds 1-4) */
        args[r].cap = cap; args[r].proof = malloc(cap);
        args[r].reps = argc > 7 ? atoi(argv[7]) : 0;
        pthread_create(&th[r], NULL, run_rank, &args[r]);
    }
    for (int r = 0; r < g_world; ++r) pthread_join(th[r], NULL);
    if (g_fail_rank >= 0) {
        /* the injected failure must end EVERY rank with an error (nobody hangs, nobody produces a proof) */
        int all = 1;
        for (int r = 0; r < g_world; ++r) { fprintf(stderr, "rank %d: %d: %s\n", r, args[r].rc, args[r].err); all = all && args[r].rc != 0; }
        printf(all ? "failure contained: every rank returned an error\n" : "failure NOT contained\n");
        return all ? 3 : 1;
    }
    for (int r = 0; r < g_world; ++r)
        if (args[r].rc) { fprintf(stderr, "rank %d: %d: %s\n", r, args[r].rc, args[r].err); return 1; }
    /* the single-GPU prover on the same trace: every rank's proof must be byte-identical to it */
    zk_ctx *ctx = NULL;
    uint8_t *one = malloc(cap), st1[32], root1[32];
    size_t len1 = 0;
    if (zk_ctx_create(0, log_n, log_b, &ctx) || zk_prove(ctx, trace, n - 1, one, cap, &len1, st1) || zk_lde(ctx) || zk_merkle_commit(ctx, 0, root1)) {
        fprintf(stderr, "%s\n", zk_last_error());
        return 1;
    }
    zk_ctx_destroy(ctx);
    for (int r = 0; r < g_world; ++r) {
        if (args[r].len != len1 || memcmp(args[r].proof, one, len1) || memcmp(args[r].state, st1, 32)) { fprintf(stderr, "rank %d: proof differs from zk_prove\n", r); return 1; }
        if (memcmp(args[r].root, root1, 32)) { fprintf(stderr, "rank %d: lde_commit root differs\n", r); return 1; }
    }
    if (zk_verify_strict(one, len1, st1, log_n, log_b, trace[n - 2])) { fprintf(stderr, "%s\n", zk_last_error()); return 1; }
    if (args[0].reps > 0) {
        /* the same domain on the one-call prover, for the ratio (total work of the sharded run / work of one proof) */
        zk_ctx *c2 = NULL;
        if (!zk_ctx_create(0, log_n, log_b, &c2) && !zk_trace_upload(c2, trace, n - 1) && !zk_prove_resident(c2, one, cap, &len1, st1)) {
            const double t0 = now_ms();
            for (int i = 0; i < args[0].reps; ++i) zk_prove_resident(c2, one, cap, &len1, st1);
            const double single = (now_ms() - t0) / args[0].reps;
            printf("timing: %d ranks sharing the GPU %.2f ms per proof (all ranks' device work, serialised) = %.2f ms per rank; one-call prover %.2f ms%s\n",
                   g_world, args[0].ms_per_proof, args[0].ms_per_proof / g_world, single, g_async ? "  [stream-ordered transport]" : "");
        }
        zk_ctx_destroy(c2);
    }
    printf("threads ok: world %d, %zu proof bytes on every rank equal zk_prove; sharded layers %u, chunked %u, board %u, all-to-all bytes per rank %.0f\n",
           g_world, len1, args[0].stats.sharded_layers, args[0].stats.chunked_layers, args[0].stats.root_board, args[0].stats.all_to_all_bytes);
    return 0;
}
