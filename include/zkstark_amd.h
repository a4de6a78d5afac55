/*
 * zkstark_amd.h -- C ABI of the MI355X-native STARK-101 prover path.
 *
 * Drop-in boundary for the hot path of Crocodoctopus/zkstark
 * (trace -> LDE -> Merkle commit -> composition -> FRI fold -> proof).  The
 * reference is a bin-only Rust crate with no FFI; the seams below are the L0
 * functions prover.rs calls.  Each entry point cites the
 * reference interface it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the Rust `extern "C"` block and the prover.rs-shaped
 * wrapper a maintainer would add.
 *
 * Conventions
 *   - every function returns int: 0 = ZK_OK, negative = error (never aborts,
 *     never throws across the boundary; the reference panics instead);
 *     zk_last_error() gives the message of the last failure on this thread;
 *   - the caller owns all host buffers; device state lives behind zk_ctx;
 *   - one context is used from one host thread at a time;
 *   - field elements cross as uint32_t canonical residues in [0, P),
 *     P = 3221225473 (main.rs:13); raw u32 challenges >= P are accepted and
 *     reduced inside (field.rs:20-24);
 *   - digests cross as 32 raw bytes in SHA-256 byte order (merkle.rs:9 Hash);
 *   - layers and trees are addressed by id: 0 = f_eval (prover.rs:70, :81),
 *     1 + r = FRI layer r (cp_evals[r] / cp_eval_merkles[r], prover.rs:192-221),
 *     r = 0 .. log_n.
 */
#ifndef ZKSTARK_AMD_H
#define ZKSTARK_AMD_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>   /* ZK_STRUCT_INIT */

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_FIELD_P 3221225473u

enum zk_status {
    ZK_OK = 0,
    ZK_ERR_INVALID = -1,  /* bad argument (size not a power of two, out of range, ...) */
    ZK_ERR_HIP = -2,      /* HIP runtime / kernel failure */
    ZK_ERR_NOMEM = -3,
    ZK_ERR_STATE = -4,    /* stage called out of order */
    ZK_ERR_BUFFER = -5,   /* caller buffer too small */
    ZK_ERR_VERIFY = -6,   /* proof rejected (reference: panic in Proof::verify) */
    ZK_ERR_CHECK = -7     /* prover self-check failed (reference: assert_eq! in prover.rs) */
};

typedef struct zk_ctx zk_ctx;
typedef struct zk_channel zk_channel;   /* Channel (channel.rs:6-37), see below */

/* Merkle hash.  SHA-256 is the reference's (merkle.rs:1-2) and the default everywhere.  The
 * field-native hash (a Poseidon2-style permutation over GF(P), csrc/fieldhash.hpp) is the build's
 * own definition for BASELINE.json configs[4]; it has no reference counterpart.  The transcript
 * (channel.rs) always uses SHA-256.  Functions with an _ex suffix take the selector. */
enum zk_hash_kind { ZK_HASH_SHA256 = 0, ZK_HASH_FIELD = 1 };

const char *zk_last_error(void);
const char *zk_version(void);

/* ---- ABI version and caller-allocated structs -------------------------------------------------------------------
 * The reference has no FFI, so the rules of this seam are the build's own:
 *   - ZK_ABI_VERSION changes whenever an exported signature or the layout of a struct below changes incompatibly; a caller
 *     compiled against this header checks zk_abi_version() == ZK_ABI_VERSION once at start-up (the C programs under examples/ and tests/ do);
 *   - every struct the CALLER allocates (zk_transcript_info, zk_kernel_stat, zk_shard_options, zk_shard_stats,
 *     zk_shard_plan_info, zk_chain_probe) starts with `uint32_t struct_size`, which the caller sets to sizeof(the struct) as
 *     its own compiler sees it (ZK_STRUCT_INIT zeroes the struct and sets it) BEFORE every call, for outputs as well as for
 *     inputs.  The library reads / writes at most struct_size bytes and refuses (ZK_ERR_INVALID, zk_last_error names the
 *     struct and both sizes) a size of 0 or one below the struct's size in ABI version 6 (the first with this rule): a
 *     caller compiled against an older, smaller layout gets an error instead of shifted fields or a stack overwrite;
 *   - new fields are appended at the END only, so a caller compiled against a smaller (>= version 6) layout keeps working:
 *     input fields it does not know are taken as 0 (= default), output fields it does not know are not written. */
#define ZK_ABI_VERSION 6u
uint32_t zk_abi_version(void);
#define ZK_STRUCT_INIT(ptr) (memset((ptr), 0, sizeof *(ptr)), (ptr)->struct_size = (uint32_t)sizeof *(ptr))
/* Hash of the sources and headers the library was built from (zkstark_amd/build.py: source_hash()); loaders
 * compare it with the tree to refuse a stale binary. */
const char *zk_build_hash(void);
/* How the host thread of the one-call provers hashes its share of the Merkle trees (the top 8 levels of every SHA-256
 * tree and the FRI layers of <= 2^9 values, csrc/host_sha.cpp; merkle.rs:14-51 is the definition either way):
 * 0 = portable code (then the provers leave every level to the device), 1 = x86 SHA extensions, 2 = SHA extensions plus
 * levels of >= 16 nodes sixteen at a time on AVX-512 registers.  Decided once from the CPU. */
int zk_host_hash_mode(void);
/* Restricts the host's hashing to `mode` (0, 1 or 2 as above; never more than the CPU has): 0 makes later contexts
 * leave every tree level to the device, 1 keeps the SHA unit but not the AVX-512 path (A/B measurements, tests).
 * Process-wide; call it while no prover is running and before the contexts it should affect are created. */
int zk_host_set_hash_mode(int mode);

/* ---- scalar field helpers on the host: field.rs:8-211 ------------------- */
uint32_t zk_field_add(uint32_t a, uint32_t b);       /* field.rs:99-111 */
uint32_t zk_field_sub(uint32_t a, uint32_t b);       /* field.rs:113-132 */
uint32_t zk_field_mul(uint32_t a, uint32_t b);       /* field.rs:134-167 */
uint32_t zk_field_neg(uint32_t a);                   /* field.rs:198-203 */
uint32_t zk_field_inv(uint32_t a);                   /* field.rs:205-210 */
uint32_t zk_field_pow(uint32_t a, uint32_t e);       /* field.rs:26-38 */
uint32_t zk_field_from_u32(uint32_t v);              /* field.rs:20-24 */
uint32_t zk_field_from_i32(int32_t v);               /* field.rs:10-18: v < 0 is -(|v| mod P) */
/* field.rs:165-177: a * b^-1.  b = 0 (mod P): the reference panics; returns 0 and sets zk_last_error. */
uint32_t zk_field_div(uint32_t a, uint32_t b);
/* field.rs:89-94 Rem<u32>: (residue of a) % rhs as a field element.  rhs = 0: the reference panics; returns 0 and
 * sets zk_last_error. */
uint32_t zk_field_rem(uint32_t a, uint32_t rhs);
uint32_t zk_field_generator(void);                   /* field.rs:52-86: the same search (first call), -> 5 */
uint32_t zk_field_root_of_unity(uint32_t log_order); /* prover.rs:48-49 */
uint32_t zk_field_order(uint32_t a);                 /* field.rs:45-49 (via P-1 = 3*2^30, not brute force) */

/* ---- context ------------------------------------------------------------- */
/* Allocates HBM for a proof with trace group size n = 2^log_n and blow-up
 * B = 2^log_blowup (reference literals: 10 and 3, prover.rs:48-57), builds the
 * twiddle and denominator tables.  One-time cost; zk_ctx_setup_ms reports it. */
int zk_ctx_create(int device, uint32_t log_n, uint32_t log_blowup, zk_ctx **out);
int zk_ctx_destroy(zk_ctx *ctx);
double zk_ctx_setup_ms(const zk_ctx *ctx);
size_t zk_ctx_device_bytes(const zk_ctx *ctx);
int zk_ctx_sync(zk_ctx *ctx);
/* Number of decommitment queries of later zk_prove* calls (1..64; default 1 = the reference,
 * prover.rs:263).  With q > 1 the q raw indices are drawn in a row and each query's openings are
 * committed in turn; q = 1 is byte-identical to the reference format. */
int zk_ctx_set_queries(zk_ctx *ctx, uint32_t n_queries);
/* Opt-in reference self-checks.  The reference asserts its way through generate_proof; with on != 0 the same
 * checkpoints run inside every later zk_prove* on the data as it sits in HBM: the interpolant passes through every
 * trace point (prover.rs:64-66), every constraint division is exact and deg cp = n - 1 (prover.rs:148-159, :169), every
 * FRI layer has its asserted degree (prover.rs:228-251).  The first checkpoint that fails is named in a ZK_ERR_CHECK
 * (zk_last_error).  Off (default): only the last-layer check of prover.rs:238 runs. */
int zk_ctx_set_checks(zk_ctx *ctx, int on);
/* Selects the Merkle hash of every later zk_merkle_commit / zk_prove* on this context. */
int zk_ctx_set_hash(zk_ctx *ctx, int hash_kind);
/* Division of the latency-bound end of zk_prove* between device and host.  A Merkle level is a chain of dependent
 * hashes (merkle.rs:40-46): ~4.6 us on a GPU wave (2 293 issue slots of 4 cycles at the ~2.1 GHz the chip holds), ~31 ns
 * per node on a CPU core with SHA extensions.  With top_log = H > 0 the device builds every SHA-256 tree with more than
 * 2^H leaves down to its 2^H nodes of depth H (a smaller tree: one level below its leaves) and the host hashes the nodes
 * above: the calling thread alone for H <= 8, a team of 2^(H-8) threads (one 256-digest sub-tree each, the workers spin
 * between the commitments of a proof) for H = 9, 10.  With tail_log = T > 0, FRI layers of <= 2^T values
 * (polynomial.rs:385 fold + merkle.rs:14 tree) are computed by the calling thread as well.  Everything the host built
 * is copied into the device arrays before the call returns, so zk_layer_read / zk_merkle_path see complete trees.
 * Default: (8, 9) when the CPU has SHA extensions, else (0, 0) = all on the device (H, T <= 10; T > 0 needs H > 0).
 * zk_merkle_commit hands over its tree top in the same way; the field hash always runs on the device.  Results are
 * identical for every setting. */
int zk_ctx_set_host_levels(zk_ctx *ctx, uint32_t top_log, uint32_t tail_log);
int zk_ctx_get_host_levels(const zk_ctx *ctx, uint32_t *top_log, uint32_t *tail_log);
/* Early launch (round 6).  A proof is a chain of commitments: digests to the host, root into the channel, challenge out, the next
 * layer's launches (prover.rs:198-225).  With on != 0, zk_prove* enqueues the fold + commit launches of the NEXT FRI round before it
 * waits for the current commitment, behind a command-processor wait on a host word (hipStreamWaitValue32), and releases them with
 * one store once the challenge is drawn: the launch call and ~2.5 us of dispatch latency per commitment leave the critical path.
 * Results are identical either way.  Off by default: measured inside one build at nothing outside the spread for a 2^24 proof,
 * -1.5 % for a 2^20 proof, -5 % at the reference's own size (profiles/r06_ab_early_launch.txt); worth switching on for small
 * domains proved one at a time.  Contexts that share a GPU (zk_prove_many) should leave it off: a stream that waits on its gate
 * holds the hardware queue another context's stream may be mapped to.  zk_ctx_get_early_launch: 1 when on AND supported. */
int zk_ctx_set_early_launch(zk_ctx *ctx, int on);
int zk_ctx_get_early_launch(const zk_ctx *ctx);
/* The HIP stream every stage is enqueued on (hipStream_t). */
void *zk_ctx_stream(zk_ctx *ctx);

/* ---- trace: prover.rs:32-42 ---------------------------------------------- */
/* a0, a1, a[i] = a[i-2]^2 + a[i-1]^2 on the host (serial recurrence). */
int zk_trace_fibsq(uint32_t a0, uint32_t a1, size_t count, uint32_t *out);
/* Uploads the n-1 trace values (prover.rs:60 interpolates exactly n-1 points). */
int zk_trace_upload(zk_ctx *ctx, const uint32_t *trace, size_t count);

/* ---- stages on context-resident data --------------------------------------- */
/* lagrange (polynomial.rs:337) + solve over w*h^i (polynomial.rs:49, prover.rs:60-70):
 * layer 0 <- f(w h^i), i < N, natural order. */
int zk_lde(zk_ctx *ctx);
/* Merkle::new over a layer (merkle.rs:14-51; prover.rs:81, :176, :214); root = node 0.  Returns once the root is known.  With
 * a host hand-over (zk_ctx_set_host_levels) the device copy of the nodes this thread hashed (the top 8 levels) is ordered on
 * the stream by the next call that reads trees or layers from the device (zk_merkle_node, zk_merkle_path, zk_layer_read,
 * zk_ctx_sync, zk_ctx_stream), not by this one: a loop of commitments does not pay a copy launch per iteration. */
int zk_merkle_commit(zk_ctx *ctx, uint32_t layer, uint8_t root_out[32]);
/* Constraint quotients and their random combination (prover.rs:101-173):
 * layer 1 <- cp(w h^i).  alpha_raw are the raw u32 challenges (prover.rs:163-165). */
int zk_compose(zk_ctx *ctx, const uint32_t alpha_raw[3]);
/* fri() + squared half domain + re-evaluation (polynomial.rs:385, prover.rs:198-211):
 * layer 2+round <- fold(layer 1+round, beta). */
int zk_fri_fold(zk_ctx *ctx, uint32_t round, uint32_t beta_raw);
/* Small device->host reads (decommit, tests). */
int zk_layer_read(zk_ctx *ctx, uint32_t layer, size_t offset, size_t count, uint32_t *out);
int zk_layer_write(zk_ctx *ctx, uint32_t layer, size_t offset, size_t count, const uint32_t *in);
/* Index<usize> for Merkle (merkle.rs:74-79). */
int zk_merkle_node(zk_ctx *ctx, uint32_t tree, size_t index, uint8_t out[32]);
/* Merkle::trace (merkle.rs:54-71): sibling of the leaf first, child of the root last.
 * out holds 32 * log2(m) bytes; *path_len receives log2(m). */
int zk_merkle_path(zk_ctx *ctx, uint32_t tree, size_t leaf, uint8_t *out, size_t *path_len);

/* ---- whole prover: generate_proof (prover.rs:9-293) ------------------------- */
/* Runs from "trace resident on device" (zk_trace_upload) to "proof bytes on host".
 * proof_out receives Channel.data (channel.rs:8), state_out Channel.state
 * (channel.rs:34-36), i.e. the two fields of Proof (proof.rs:5-8). */
int zk_prove_resident(zk_ctx *ctx, uint8_t *proof_out, size_t cap, size_t *proof_len,
                      uint8_t state_out[32]);
/* generate_proof(channel: Channel) -> Proof (prover.rs:9) literally: proves the resident trace on the CALLER'S
 * channel (channel.rs:6-37).  Every commitment is appended to `ch` and every challenge is drawn from it, so a
 * channel that already holds a transcript prefix yields the proof bound to that prefix; Proof{state, data}
 * (proof.rs:5-8, channel.rs:34-36) is then zk_channel_state / zk_channel_data.  With a fresh channel
 * (main.rs:19) the bytes equal zk_prove_resident's. */
int zk_prove_channel(zk_ctx *ctx, zk_channel *ch);
/* zk_prove_resident on `count` (1..16) distinct contexts at once, one host thread each: the latency-bound
 * phases of one proof overlap the hashing of the others on the same GPU.  Proof i goes to
 * proofs_out + i*stride (stride >= the proof length), its length to lens_out[i], its state to
 * states_out + 32*i. */
int zk_prove_many(zk_ctx *const *ctxs, size_t count, uint8_t *proofs_out, size_t stride, size_t *lens_out,
                  uint8_t *states_out);
/* zk_trace_upload + zk_prove_resident. */
int zk_prove(zk_ctx *ctx, const uint32_t *trace, size_t count, uint8_t *proof_out, size_t cap,
             size_t *proof_len, uint8_t state_out[32]);
/* Challenges and checkpoints of the last zk_prove* on this context (tests/diagnostics). */
typedef struct zk_transcript_info {
    uint32_t struct_size;   /* sizeof(zk_transcript_info), set by the caller (see "ABI version" above) */
    uint32_t alpha_raw[3];
    uint32_t beta_raw[32];
    uint32_t free_term;
    uint32_t query_raw;
    uint32_t public_last; /* a[n-2] */
    uint8_t roots[34][32]; /* trees 0 .. log_n + 1 */
} zk_transcript_info;
int zk_last_transcript(const zk_ctx *ctx, zk_transcript_info *out);
/* Optional per-kernel timing with HIP events on the context stream.  class_mask selects
 * the kernel classes to bracket (bit i = class i below; 0 = off, the default), so a
 * benchmark can time only the dominant kernel inside its timed region. */
enum zk_kernel_class {
    ZK_K_NTT = 0,           /* ntt_pass_kernel (iNTT / LDE passes) */
    ZK_K_MERKLE_LEAF = 1,   /* merkle_subtree_kernel<leaf>: leaf hashes + k inner levels */
    ZK_K_MERKLE_INNER = 2,  /* merkle_subtree_kernel<inner> */
    ZK_K_MERKLE_TOP = 3,    /* merkle_wg_kernel: the latency-bound levels, workgroup-local */
    ZK_K_COMPOSE = 4,
    ZK_K_FOLD = 5,
    ZK_K_GATHER = 6,
    ZK_K_COUNT = 7
};
typedef struct zk_kernel_stat {
    uint32_t struct_size;   /* sizeof(zk_kernel_stat), set by the caller in out[0]: the stride of the array */
    uint32_t reserved;
    uint64_t launches;
    double ms;     /* summed launch durations */
    double bytes;  /* summed ALGORITHMIC bytes of those launches */
    double ops;    /* summed compulsory 32-bit VALU lane-ops (SHA-256 kernels; 0 elsewhere) */
} zk_kernel_stat;
int zk_ctx_set_profiling(zk_ctx *ctx, uint32_t class_mask);
/* Copies the accumulated statistics (count <= ZK_K_COUNT entries); reset != 0 clears them. */
int zk_kernel_stats(zk_ctx *ctx, zk_kernel_stat *out, size_t count, int reset);

/* ---- batched proving ---------------------------------
 * 2^log_batch independent proofs of one size (prover.rs:9-293 each, its own Channel each) computed in
 * lockstep: the layers of the batch are stored proof-major, so every stage is ONE launch of the kernels
 * a single proof uses on a domain batch times larger, and the trees of the batch are the bottom of one
 * heap whose nodes of depth log_batch are the per-proof roots.  Every proof is byte-identical to what
 * zk_prove returns for the same trace.  log_batch <= 10; (log_n, log_blowup) as for zk_ctx_create (log_n >= 2, != 3), so
 * every proof a batch produces can be checked by zk_verify*. */
typedef struct zk_batch zk_batch;
int zk_batch_create(int device, uint32_t log_n, uint32_t log_blowup, uint32_t log_batch, zk_batch **out);
int zk_batch_destroy(zk_batch *b);
size_t zk_batch_size(const zk_batch *b);                    /* 2^log_batch */
/* As zk_ctx_set_queries (1..16 here) and zk_ctx_set_hash, for every proof of the batch. */
int zk_batch_set_queries(zk_batch *b, uint32_t n_queries);
int zk_batch_set_hash(zk_batch *b, int hash_kind);
/* on = 0: every tree level of the batch on the device (default: the host threads hash the top levels of each proof's
 * trees when the CPU has SHA extensions, as zk_ctx_set_host_levels).  Results are identical. */
int zk_batch_set_host_levels(zk_batch *b, int on);
/* Host threads that run the per-proof transcript steps and decommit hashing (default: hardware threads, at most 16).
 * Like every zk_batch_set_* / trace call it is refused with ZK_ERR_STATE while a zk_batch_prove runs on the batch (one batch is
 * used from one host thread at a time; the guard turns a misuse into an error instead of a freed pool under a running proof). */
int zk_batch_set_threads(zk_batch *b, uint32_t threads);
size_t zk_batch_device_bytes(const zk_batch *b);
/* traces: [batch][n-1] canonical residues (prover.rs:32-39 per proof), uploaded and kept resident. */
int zk_batch_set_traces(zk_batch *b, const uint32_t *traces);
/* The same traces generated on the device from seeds a0[p], a1[p] (one lane per trace). */
int zk_batch_gen_fibsq(zk_batch *b, const uint32_t *a0, const uint32_t *a1);
/* out[p] = a[n-2] of proof p: the public input its verifier needs (prover.rs:42, proof.rs:68). */
int zk_batch_public_last(const zk_batch *b, uint32_t *out);
/* proofs_out: [batch][stride] bytes, stride >= zk_proof_data_len(log_n, log_blowup); states_out:
 * [batch][32] (with q queries: zk_proof_data_len_queries).  Fails with ZK_ERR_CHECK, naming the proof, if a trace breaks the constraints. */
int zk_batch_prove(zk_batch *b, uint8_t *proofs_out, size_t stride, uint8_t *states_out);

/* ---- proof: proof.rs ------------------------------------------------------- */
/* Proof::verify (proof.rs:15-149), CPU only, generalised from the literals
 * (1024, 8192, 10, 2338775057) to (log_n, log_blowup, public_last). */
int zk_verify(const uint8_t *proof, size_t len, uint32_t log_n, uint32_t log_blowup,
              uint32_t public_last);
int zk_verify_ex(const uint8_t *proof, size_t len, uint32_t log_n, uint32_t log_blowup,
                 uint32_t public_last, int hash_kind);
/* General form: hash selector, q queries, and (state != NULL) the transcript replay of zk_verify_strict. */
int zk_verify_queries(const uint8_t *proof, size_t len, const uint8_t *state, uint32_t log_n, uint32_t log_blowup,
                      uint32_t public_last, int hash_kind, uint32_t n_queries);
/* zk_verify plus a replay of the Fiat-Shamir channel over the proof bytes: every challenge must be
 * the one the transcript yields at that point and `state` (Proof.state, proof.rs:6, which the
 * reference stores but never checks) must be the final channel state.  SURVEY.md section 8f item 1. */
int zk_verify_strict(const uint8_t *proof, size_t len, const uint8_t state[32], uint32_t log_n,
                     uint32_t log_blowup, uint32_t public_last);
/* Proof::size (proof.rs:151-154). */
size_t zk_proof_size(size_t data_len);
size_t zk_proof_data_len(uint32_t log_n, uint32_t log_blowup);
size_t zk_proof_data_len_queries(uint32_t log_n, uint32_t log_blowup, uint32_t n_queries);
/* compute_root_from_path (merkle.rs:82-110), CPU. */
int zk_compute_root_from_path(uint32_t element, size_t index, const uint8_t *path, size_t path_len,
                              uint8_t out[32]);

int zk_compute_root_from_path_ex(uint32_t element, size_t index, const uint8_t *path, size_t path_len,
                                 uint8_t out[32], int hash_kind);

/* ---- Channel (channel.rs:6-37), host only ------------------------------------ */
int zk_channel_new(zk_channel **out);                                        /* channel.rs:12 */
int zk_channel_free(zk_channel *ch);
/* Adopts the two fields of a Channel kept on the caller's side (channel.rs:6-9: state, data): how the reference's
 * Rust Channel argument of generate_proof (prover.rs:9) crosses the FFI; read back with zk_channel_state / _data. */
int zk_channel_import(zk_channel *ch, const uint8_t state[32], const uint8_t *data, size_t n);
int zk_channel_commit(zk_channel *ch, const uint8_t *bytes, size_t n);       /* channel.rs:19 */
int zk_channel_get_u32(zk_channel *ch, uint32_t *out);                       /* channel.rs:28 */
int zk_channel_state(const zk_channel *ch, uint8_t out[32]);
size_t zk_channel_data_len(const zk_channel *ch);
int zk_channel_data(const zk_channel *ch, uint8_t *out, size_t cap);         /* channel.rs:34 */

/* FRI tail: the last layers of a proof whose earlier layers live elsewhere (one proof sharded over
 * several GPUs hands over once a layer is small enough to be replicated).  Layer rho0 of a
 * (log_n, log_blowup) proof is layer 0 of a domain with log_n_tail = log_n - rho0 and shift
 * w^(2^rho0).  zk_tail_run copies the handed-over layer (2^(log_n_tail+log_blowup) words at d_layer0,
 * produced on src_stream), commits it, then runs the remaining log_n_tail rounds of prover.rs:198-225
 * (fused fold + commit) on the caller's channel: betas_out[log_n_tail], roots_out[(log_n_tail+1)*32].
 * zk_tail_open gathers the openings of prover.rs:280-289 for the tail layers.  Destroy with
 * zk_ctx_destroy. */
int zk_tail_create(int device, uint32_t log_n_tail, uint32_t log_blowup, uint32_t shift, zk_ctx **out);
int zk_tail_run(zk_ctx *tail, const uint32_t *d_layer0, void *src_stream, zk_channel *ch, int hash_kind,
                uint32_t *betas_out, uint8_t *roots_out, uint32_t *free_term_out);
int zk_tail_open(zk_ctx *tail, size_t x, uint32_t *vals_out, uint8_t *paths_out);

/* ---- one proof sharded over the GPUs of one node (BASELINE.json configs[3]) ----------------
 * generate_proof (prover.rs:9-293) with the evaluation domain distributed CYCLICALLY over `world` ranks, one process
 * (or thread) per GPU: rank r holds the elements i = r (mod world) of every layer, which is again a coset domain, so
 * LDE, composition and every fold run the single-GPU kernels with no communication.  The only exchange is the
 * commitment: one all-to-all per committed layer turns the cyclic layout into contiguous leaf blocks (the transpose
 * of a four-step NTT over the ranks), each rank hashes its subtree, the `world` subtree roots are exchanged and the
 * top log2(world) levels are hashed on the host.  Layers below 2^21 values (2^20 from 4 ranks on) are replicated and finished by every rank
 * (zk_tail_*).  Every rank runs the same transcript and returns the same proof bytes, identical to zk_prove's.
 *
 * Transport.  By default the collectives are RCCL's (librccl.so.1 is loaded at run time; grouped ncclSend/ncclRecv for
 * the all-to-all, ncclAllGather): rank 0 calls zk_shard_unique_id, the caller distributes the 128 bytes (any side
 * channel: MPI, a file, torch.distributed) and every rank passes them to zk_shard_create.  A caller that owns its own
 * communication passes a zk_shard_transport instead (device pointers, stream-ordered on `stream`); tests use that to
 * run several ranks on one GPU.  world must be a power of two dividing the blow-up. */
typedef struct zk_shard zk_shard;
#define ZK_SHARD_ID_BYTES 128
typedef struct zk_shard_transport {
    void *user;
    /* send[p] (words u32) goes to rank p, recv[q] (words u32) comes from rank q; p, q < world, own part included */
    int (*all_to_all)(void *user, const uint32_t *const *send, uint32_t *const *recv, size_t words, void *stream);
    /* recv[q * words ..] = rank q's send */
    int (*all_gather)(void *user, const uint32_t *send, uint32_t *recv, size_t words, void *stream);
} zk_shard_transport;
typedef struct zk_shard_options {   /* zero = default (struct_size excepted) */
    uint32_t struct_size;       /* sizeof(zk_shard_options), set by the caller (see "ABI version" above) */
    uint32_t min_layer_log;     /* a FRI layer stays sharded while it has >= 2^this values in total (21; 20 from 4 ranks on) */
    uint32_t min_chunk_log;     /* ... and >= 2^this leaves per (rank, peer) piece (14) */
    uint32_t overlap_min_log;   /* pieces of >= 2^this words are exchanged in 4 chunks overlapped with the hashing (21) */
    int force_collectives;      /* run the collectives even with world = 1 (exercises the transport on one GPU) */
    int no_root_board;          /* exchange subtree roots with an all-gather instead of the shared-memory board */
    int plain_collectives;      /* no chunked exchange and no shared-memory roots: one all-to-all per layer on the main
                                   stream, subtree roots by all-gather (the fall-back rung of bench.py --gpus N) */
    int single_build_stream;    /* chunk builds of a layer on one stream (A/B; default: two alternating streams) */
    int single_communicator;    /* built-in RCCL transport: the exchange stream shares the main communicator (default: the
                                   chunked exchanges get a communicator of their own, see below) */
    int exchange_cp;            /* 1: commit cp by exchanging it like every other layer (what rounds 1-4 did; A/B).  Default:
                                   cp over a rank's block is recomputed from the block of f the rank received for the
                                   commitment of f, inside the leaf hashing -- no exchange for cp (zk_shard_plan_info.cp_from_f) */
    double timeout_s;           /* bound of every host-side wait on a peer; 0 = environment ZK_SHARD_TIMEOUT_S, else 120 s */
    int peer_copy;              /* transport == NULL: instead of RCCL, the built-in PEER-COPY transport (csrc/peer.hpp): every rank of
                                   the node publishes the IPC handle of ONE staging buffer on a shared-memory page at creation, a
                                   collective copies its pieces there and every rank pulls its piece with a device-to-device copy.
                                   Host-synchronous (implies plain_collectives); `id` is any 128 bytes shared by the ranks.  The rung
                                   below "RCCL, plain collectives": for a node where no communicator can be formed (bench.py --gpus N
                                   falls back to it) */
    int reserved;
} zk_shard_options;
typedef struct zk_shard_stats {
    uint32_t struct_size;       /* sizeof(zk_shard_stats), set by the caller (see "ABI version" above) */
    uint32_t sharded_layers;    /* FRI layers 0 .. sharded_layers-1 (and f) are distributed; the rest is the replicated tail */
    uint32_t root_board;        /* 1: subtree roots travel through shared memory (all ranks on one node) */
    uint32_t chunked_layers;    /* layers of the last proof exchanged in chunks overlapped with the hashing */
    uint32_t native_rccl;       /* 1: the built-in RCCL transport */
    uint32_t rccl_nranks;       /* ncclCommCount of the communicator (checked against `world` at creation); 0 without RCCL */
    uint32_t communicators;     /* RCCL communicators in use: 2 when the chunked exchanges have their own, else 1; 0 without RCCL */
    double sent_bytes;          /* bytes this rank sent to OTHER ranks during the last zk_shard_prove* / lde_commit */
    double all_to_all_bytes;    /* ... of which in the per-commitment all-to-alls */
    double setup_ms;
    double device_bytes;
    /* zk_shard_set_profiling(s, 1): HIP events around the exchanges of the last zk_shard_prove* / lde_commit (0 otherwise).
     * exchange_ms: summed durations of the all-to-alls and all-gathers on their streams (for a chunked layer this includes
     * the time the exchange kernel waits for compute units beside the hashing); exposed_exchange_ms: the part of it the
     * hashing streams spent stalled (a plain exchange is exposed in full, a chunk only while its build waits for it);
     * tail_ms: host time of the replicated tail (all-gather of the first replicated layer, then zk_tail_run);
     * decommit_ms: host time from the query index to the last opening in the channel; selftest_ms: the known-pattern
     * exchange run by zk_shard_create. */
    double exchange_ms;
    double exposed_exchange_ms;
    double tail_ms;
    double selftest_ms;
    double decommit_ms;         /* host time of the decommitment (prover.rs:266-289), all queries, profiling on */
    uint32_t exchanges;         /* collectives timed (profiling on) */
    uint32_t selftest_ok;       /* 1: zk_shard_create's known-pattern all-to-all + all-gather arrived in the right places */
    uint32_t peer_copy;         /* 1: the built-in peer-copy transport (zk_shard_options.peer_copy) */
    uint32_t reserved;
} zk_shard_stats;
/* The layout of a sharded proof, as a pure function of its arguments (no GPU, no communication): what
 * zk_shard_create will do.  Layer ids: 0 = f_eval, 1 + r = FRI layer r. */
typedef struct zk_shard_plan_info {
    uint32_t struct_size;        /* sizeof(zk_shard_plan_info), set by the caller (see "ABI version" above) */
    uint32_t world;
    uint32_t log_world;
    uint32_t sharded_layers;     /* FRI layers 0 .. sharded_layers-1 (and f) are distributed */
    uint32_t tail_rounds;        /* FRI rounds of the replicated tail */
    uint32_t chunked_layers;     /* committed layers exchanged in chunks overlapped with the hashing */
    uint32_t chunked_mask;       /* bit id set: layer id is exchanged in chunks */
    uint32_t log_chunks;         /* a chunked layer is exchanged in 2^log_chunks chunks */
    uint32_t min_layer_log;      /* the thresholds in force (options and defaults; plain_collectives: overlap_min_log = 99) */
    uint32_t min_chunk_log;
    uint32_t overlap_min_log;
    uint32_t piece_log[32];      /* log2 words of one (rank, peer) piece of layer id */
    double all_to_all_bytes;     /* bytes one rank sends to its peers in the all-to-alls of one proof */
    double lde_commit_bytes;     /* ... of zk_shard_lde_commit (layer 0 only) */
    uint32_t cp_from_f;          /* 1: cp (layer id 1) is committed without an exchange: recomputed over the rank's block from the
                                    received block of f plus a 2B-word all-gather (the positions after the block); its piece
                                    is then not part of all_to_all_bytes */
    uint32_t reserved;
} zk_shard_plan_info;
int zk_shard_plan(int world, uint32_t log_n, uint32_t log_blowup, const zk_shard_options *opt, zk_shard_plan_info *out);
/* Streams and communicators.  Plain exchanges run on the prover's main stream.  Chunked exchanges (pieces of >=
 * 2^overlap_min_log words) run on a second, high-priority stream beside the hashing; with the built-in transport they use a
 * second RCCL communicator created for that stream (rank 0 draws its id and the first communicator distributes it), so no
 * communicator is ever driven from two streams, and every collective additionally waits (HIP event) for the previous
 * collective of the other stream, so no ordering rests on RCCL's internal serialisation.  A caller transport is called
 * with either stream and must tolerate that.
 * zk_shard_create ends with a self-test when collectives are in use: an all-to-all (on each stream in use) and an
 * all-gather (on the main stream, and on the exchange stream when the halo of cp travels there) of a known pattern, the
 * all-to-all at the size of the largest piece; every word must arrive at its place, else the call
 * fails with ZK_ERR_HIP naming the first wrong (peer, word).  It also brings RCCL's lazy connections up before the
 * first proof.  zk_shard_self_test repeats it on request.
 * ncclCommInitRank itself is not bounded by this library (it cannot be cancelled): a caller that must not hang runs
 * its ranks under a watchdog (bench.py does).
 *
 * ncclGetUniqueId through the same run-time loaded RCCL (rank 0, before zk_shard_create). */
int zk_shard_unique_id(uint8_t id_out[ZK_SHARD_ID_BYTES]);
/* Collective over the `world` ranks (ncclCommInitRank when transport is NULL).  id: the shared 128 bytes; with a
 * caller transport they only name the shared-memory root board (NULL: no board).  opt may be NULL. */
int zk_shard_create(int device, int rank, int world, const uint8_t *id, const zk_shard_transport *transport,
                    const zk_shard_options *opt, uint32_t log_n, uint32_t log_blowup, zk_shard **out);
int zk_shard_destroy(zk_shard *s);
/* The same n-1 trace values on every rank (prover.rs:32-39; the interpolant is replicated). */
int zk_shard_trace_upload(zk_shard *s, const uint32_t *trace, size_t count);
/* generate_proof on the caller's channel (prover.rs:9), collectively; every rank gets the same transcript. */
int zk_shard_prove_channel(zk_shard *s, zk_channel *ch);
int zk_shard_prove(zk_shard *s, uint8_t *proof_out, size_t cap, size_t *proof_len, uint8_t state_out[32]);
/* configs[3] shape: sharded LDE, all-to-all transpose, Merkle commit; root of f_eval (prover.rs:60-85). */
int zk_shard_lde_commit(zk_shard *s, uint8_t root_out[32]);
/* Merkle hash (zk_hash_kind) and number of decommitment queries, as zk_ctx_set_hash / zk_ctx_set_queries; every rank
 * must use the same values.  Verify with zk_verify_queries(..., hash_kind, n_queries). */
int zk_shard_set_hash(zk_shard *s, int hash_kind);
int zk_shard_set_queries(zk_shard *s, uint32_t n_queries);
/* Failure handling.  A rank that leaves zk_shard_prove* / zk_shard_lde_commit with an error posts an abort on the shared
 * root board (peers waiting for it return ZK_ERR_HIP naming this rank instead of waiting out zk_shard_options.timeout_s),
 * refuses further proofs, and zk_shard_destroy then aborts its RCCL communicators (ncclCommAbort).  The board's abort
 * words are mapped whenever the ranks share a node, also when the roots themselves travel by all-gather
 * (no_root_board, plain_collectives).  With a caller transport nothing can abort a collective that is already
 * enqueued: zk_shard_destroy of a failed prover then does not wait for its streams (they are leaked with a message).
 * zk_shard_inject_failure makes a rank fail that way on purpose (tests). */
int zk_shard_inject_failure(zk_shard *s, int code);
/* Known-pattern all-to-all + all-gather through the prover's transport (collective; see above).  The pattern is written into
 * the storage of layer 0 (f_eval) and the receive buffer: the layers resident from an earlier zk_shard_prove* / _lde_commit are
 * INVALID afterwards (zk_shard_layer_read(0, ..) would return pattern words) until the next proof recomputes them. */
int zk_shard_self_test(zk_shard *s);
/* on != 0: time the exchanges of later proofs with HIP events (zk_shard_stats.exchange_ms ...); costs a few event
 * records per layer, so benchmarks switch it on for one untimed proof. */
int zk_shard_set_profiling(zk_shard *s, int on);
int zk_shard_last_transcript(const zk_shard *s, zk_transcript_info *out);
/* This rank's shard of a layer (0 = f_eval, 1 + r = FRI layer r < sharded_layers): element j is global index
 * rank + world * j. */
int zk_shard_layer_read(zk_shard *s, uint32_t layer, size_t offset, size_t count, uint32_t *out);
int zk_shard_get_stats(const zk_shard *s, zk_shard_stats *out);

/* Timing of the zk_dev_* launches (process-wide; same classes and semantics as
 * zk_ctx_set_profiling / zk_kernel_stats). */
int zk_dev_set_profiling(uint32_t class_mask);
int zk_dev_kernel_stats(zk_kernel_stat *out, size_t count, int reset);

/* Roofline probe (measurement only; bench.py `roofline.valu.chain_*`): every lane of waves_per_simd resident waves per
 * SIMD runs a dependent chain of `hashes` inner hashes (merkle.rs:42-45 shape) with no memory traffic, `launches`
 * launches back to back.  ns_per_hash_per_simd is the steady-state time one SIMD needs per hash of one wave; divided
 * by the hash's VALU instruction count it is the issue rate the Merkle kernels can reach at best at that residency. */
typedef struct zk_chain_probe {
    uint32_t struct_size;          /* sizeof(zk_chain_probe), set by the caller (see "ABI version" above) */
    uint32_t reserved;
    double ns_per_hash_per_simd;
    double clock_ghz;              /* median over the waves: shader clocks per 100 MHz reference tick */
    double ms;                     /* wall time of the timed launches (HIP events) */
    uint32_t waves_per_simd;
    uint32_t launches;
    uint32_t hashes;
    uint32_t cus;                  /* compute units of the device */
} zk_chain_probe;
int zk_probe_hash_chain(int device, int hash_kind, uint32_t waves_per_simd, uint32_t hashes, uint32_t launches, zk_chain_probe *out);

/* Test hook (configs[4]): the field hash exists in several forms on the device -- double precision one lane per node (every tree is
 * built with it, csrc/fieldhash_f64.hpp), its quad and 16-lane row forms (the narrow tree levels), 32-bit Montgomery (the host's form:
 * verifier, tree tops of the sharded prover) and the 32-bit row form of rounds 3-4.  Hashes `count` pseudo-random inputs, every 16th an edge pattern (words 0, P - 1, raw words >= P), through
 * all of them: *mismatches = inputs on which they differ or a digest word is not canonical (must be 0). */
int zk_probe_fieldhash_forms(int device, uint32_t count, uint32_t seed, uint32_t *mismatches, uint32_t *first_bad);

/* ---- stand-alone primitives on host buffers (upload, run on the GPU, download) -- */
/* Merkle::new(size, data) (merkle.rs:14): nodes_out = (2m-1)*32 bytes, heap order. */
int zk_merkle_build_host(int device, const uint32_t *vals, size_t m, uint8_t *nodes_out);
int zk_merkle_build_host_ex(int device, const uint32_t *vals, size_t m, uint8_t *nodes_out, int hash_kind);
/* Natural-order NTT / inverse NTT of size 2^log_m with the canonical root
 * zk_field_root_of_unity(log_m); in place on the host buffer. */
int zk_ntt_host(int device, uint32_t *data, uint32_t log_m, int inverse);
/* trace (n-1 values) -> N coset evaluations (same as zk_trace_upload + zk_lde + read). */
int zk_lde_host(int device, const uint32_t *trace, uint32_t log_n, uint32_t log_blowup, uint32_t *out);

/* ---- coset domains and device-pointer primitives (stream-ordered) ------------------
 * For callers that own the device buffers and the orchestration (the sharded prover, csrc/shard.hip,
 * drives these between RCCL collectives).  A domain is
 * {shift * h^i, i < 2^(log_n+log_blowup)}, h = zk_field_root_of_unity(log_n+log_blowup), with
 * the tables the kernels need; the reference's domain is shift = 5 (prover.rs:69).  log_blowup
 * may be 0.  fold_only != 0 builds only what zk_dev_fri_fold needs.  Entry points that take only device
 * pointers and a stream (zk_dev_merkle_build*, zk_dev_interleave, zk_dev_gather, ...) launch on the
 * calling thread's current device: select the device the buffers live on first (hipSetDevice /
 * torch.cuda.set_device), as one process per GPU does once at start-up. */
typedef struct zk_dom zk_dom;
int zk_dom_create(int device, uint32_t log_n, uint32_t log_blowup, uint32_t shift, int fold_only, zk_dom **out);
int zk_dom_destroy(zk_dom *dom);
/* lagrange + solve (polynomial.rs:337, :49; prover.rs:60-70).  d_trace: n words = a[0..n-2] followed
 * by 0; d_coef: 2n words scratch; d_out: N words, natural order. */
int zk_dev_lde(const zk_dom *dom, const uint32_t *d_trace, uint32_t *d_coef, uint32_t *d_out, void *stream);
/* prover.rs:101-173 pointwise; first = a[0], last = a[n-2]. */
int zk_dev_compose(const zk_dom *dom, const uint32_t *d_f, uint32_t *d_cp, uint32_t first, uint32_t last,
                   const uint32_t alpha_raw[3], void *stream);
/* polynomial.rs:385 + prover.rs:204-211: 2^log_m values of FRI layer `round` -> 2^(log_m-1). */
int zk_dev_fri_fold(const zk_dom *dom, const uint32_t *d_in, uint32_t *d_out, uint32_t log_m, uint32_t round,
                    uint32_t beta_raw, void *stream);
/* Batch trace generation: prover.rs:32-39 is serial per trace, so one
 * lane generates one trace; out[t*count + i] = a_i of trace t seeded by (a0[t], a1[t]). */
int zk_dev_trace_fibsq_batch(const uint32_t *d_a0, const uint32_t *d_a1, uint32_t batch, uint32_t count,
                             uint32_t *d_out, void *stream);
int zk_trace_fibsq_batch_host(int device, const uint32_t *a0, const uint32_t *a1, uint32_t batch,
                              uint32_t count, uint32_t *out);
/* out[u * parts + q] = in[q * cnt + u]: cyclic <-> block layout around the all-to-all. */
int zk_dev_interleave(const uint32_t *d_in, uint32_t *d_out, uint32_t log_parts, uint32_t log_cnt, void *stream);
/* d_out[i*words ..] = d_src[d_offsets[i] .. + words]. */
int zk_dev_gather(const uint32_t *d_src, const uint64_t *d_offsets, uint32_t count, uint32_t words,
                  uint32_t *d_out, void *stream);
/* d_vals: m u32 on the device; d_nodes: (2m-1)*8 u32 state words, heap order. */
int zk_dev_merkle_build(const uint32_t *d_vals, uint32_t log_m, uint32_t *d_nodes, void *stream);
int zk_dev_merkle_build_ex(const uint32_t *d_vals, uint32_t log_m, uint32_t *d_nodes, void *stream, int hash_kind);
/* Same tree, with the leaves still in all-to-all order: 2^log_parts pieces of 2^log_cnt words, leaf
 * u*parts + q = d_recv[q*cnt + u] (zk_dev_interleave fused into the leaf hashing). */
int zk_dev_merkle_build_interleaved(const uint32_t *d_recv, uint32_t log_parts, uint32_t log_cnt,
                                    uint32_t *d_nodes, void *stream, int hash_kind);
/* Commitment with the root handed to the host (what prover.rs:85 feeds the channel): the tree of
 * zk_dev_merkle_build_ex (log_parts = 0, d_src in natural order) or zk_dev_merkle_build_interleaved
 * (log_parts > 0).  With SHA-256 and SHA extensions on the CPU the device stops at depth 8, the calling
 * thread hashes the 255 nodes above (see zk_ctx_set_host_levels) and a copy ordered on `stream` completes
 * d_nodes; the call returns once the root is known.  One committer per host thread;
 * consecutive commits may use different streams (the staging buffer of the previous commit is released by an event). */
typedef struct zk_committer zk_committer;
int zk_committer_create(int device, zk_committer **out);
int zk_committer_destroy(zk_committer *k);
/* Hand-over depth of later commits (top_log of zk_ctx_set_host_levels; <= 8; 0 = the device builds to the root). */
int zk_committer_set_top(zk_committer *k, uint32_t top_log);
int zk_dev_merkle_commit(zk_committer *k, const uint32_t *d_src, uint32_t log_parts, uint32_t log_cnt,
                         uint32_t *d_nodes, void *stream, int hash_kind, uint8_t root_out[32]);
/* zk_dev_merkle_finish with the same hand-over: for a tree built with zk_dev_merkle_build_chunk. */
int zk_dev_merkle_commit_finish(zk_committer *k, uint32_t *d_nodes, uint32_t log_m, uint32_t log_chunks,
                                void *stream, int hash_kind, uint8_t root_out[32]);
/* The same tree in 2^c aligned chunks, so that hashing chunk i overlaps the exchange of chunk i+1: chunk
 * `chunk` covers leaves [chunk << s, (chunk+1) << s), s = log_parts + log_cnt, arriving in its own
 * receive buffer in all-to-all order; the throughput-bound levels are built in place in the heap over
 * 2^log_m leaves.  zk_dev_merkle_finish then runs the latency-bound top of the whole tree once. */
int zk_dev_merkle_build_chunk(const uint32_t *d_recv, uint32_t log_parts, uint32_t log_cnt, uint32_t *d_nodes,
                              uint32_t log_m, uint32_t chunk, void *stream, int hash_kind);
int zk_dev_merkle_finish(uint32_t *d_nodes, uint32_t log_m, uint32_t log_chunks, void *stream, int hash_kind);
/* Tuning / test hook, process-wide: a Merkle build runs throughput launches (a wave owns 64 * 2^k leaves) while a level
 * has more than 2^log_nodes nodes and finishes the tree in workgroup-local launches below that (default 17: measured
 * flat between 16 and 18; 0 restores the default; 12 .. 24).  Results do not depend on it; tests lower it to drive
 * small trees through the chunked build (zk_dev_merkle_build_chunk / _finish).  Call it while no build is in flight. */
int zk_dev_set_merkle_latency_log(uint32_t log_nodes);
/* Byte view of nodes stored as state words: out[32] for node `index`. */
int zk_dev_merkle_node(const uint32_t *d_nodes, size_t index, uint8_t out[32], void *stream);

#ifdef __cplusplus
}
#endif
#endif
