/*
 * stark101_oracle.h -- CPU oracle for the STARK-101 prover hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm in
 * Crocodoctopus/zkstark (reference files cited per function, paths relative to
 * /root/reference).  Only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py may load it; the product library (zkstark_amd/) never links,
 * loads or calls anything in oracle/.
 *
 * Parity pin: the reference is a Rust bin crate that cannot be built here (no
 * cargo/rustc, crates not vendored), so the oracle is pinned by every known
 * answer the reference itself asserts: the five unit
 * tests (field.rs:213, polynomial.rs:402/428/456, merkle.rs:112) and the
 * assert_eq! checkpoints in prover.rs:42-159, 169, 228-251.  The transcript
 * byte encoding (bincode 1.x defaults) has no golden vector in the reference:
 * "parity unpinned" for proof bytes/challenges; everything upstream of the
 * transcript (field values, Merkle nodes) is pinned.
 *
 * Two modes are provided for the polynomial work:
 *   ORC_MODE_NAIVE  follows polynomial.rs literally (Lagrange through ll/lr
 *                   products, per-term pow in solve, recursive long division,
 *                   coefficient-form FRI fold + re-evaluation).  O(n^3).
 *   ORC_MODE_NTT    the size-generic O(N log N) restatement (iNTT + coset NTT,
 *                   pointwise composition, evaluation-form fold).  Both must
 *                   agree bit for bit at n = 1024.
 */
#ifndef STARK101_ORACLE_H
#define STARK101_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_P 3221225473u /* main.rs:13 */

enum { ORC_MODE_NTT = 0, ORC_MODE_NAIVE = 1 };

/* ---- field.rs ---------------------------------------------------------- */
uint32_t orc_add(uint32_t a, uint32_t b, uint32_t p);
uint32_t orc_sub(uint32_t a, uint32_t b, uint32_t p);
uint32_t orc_mul(uint32_t a, uint32_t b, uint32_t p);
uint32_t orc_neg(uint32_t a, uint32_t p);
uint32_t orc_pow(uint32_t a, uint32_t e, uint32_t p);
uint32_t orc_inv(uint32_t a, uint32_t p);
uint32_t orc_from_u32(uint32_t v, uint32_t p);  /* field.rs:20-24 */
uint32_t orc_from_i32(int32_t v, uint32_t p);   /* field.rs:10-18 */
uint32_t orc_div(uint32_t a, uint32_t b, uint32_t p);      /* field.rs:165-177 */
uint32_t orc_rem(uint32_t a, uint32_t rhs, uint32_t p);    /* field.rs:89-94 */
uint32_t orc_order(uint32_t a, uint32_t p);     /* field.rs:45-49 */
uint32_t orc_generator(uint32_t p);             /* field.rs:52-86 */

/* ---- polynomial.rs (coefficients LOW degree first, as stored at :31) ---- */
/* lagrange(): npts points -> npts coefficients (top ones may be zero). */
void orc_lagrange_naive(const uint32_t *xs, const uint32_t *ys, size_t npts,
                        uint32_t *coef_out, uint32_t p);
uint32_t orc_poly_solve_naive(const uint32_t *coef, size_t len, uint32_t x, uint32_t p);
/* div(): returns quotient length; *rem_len = remainder length (0 = exact). */
size_t orc_poly_div(const uint32_t *num, size_t num_len, const uint32_t *den, size_t den_len,
                    uint32_t *quot_out, uint32_t *rem_out, size_t *rem_len, uint32_t p);
/* i32 variant pinned by div_test (polynomial.rs:456-490). */
size_t orc_poly_div_i32(const int32_t *num, size_t num_len, const int32_t *den, size_t den_len,
                        int32_t *quot_out, int32_t *rem_out, size_t *rem_len);
/* fri(): out_len = len/2 (polynomial.rs:385-400). */
void orc_fri_coef_fold(const uint32_t *coef, size_t len, uint32_t beta, uint32_t *out, uint32_t p);

/* ---- size-generic transforms (the build's restatement) --- */
void orc_set_threads(int nthreads);
/* In-place radix-2 transform, natural order in and out.  root must have order 2^log_n. */
void orc_ntt(uint32_t *data, uint32_t log_n, uint32_t root);
void orc_intt(uint32_t *data, uint32_t log_n, uint32_t root);
/* Virtual last trace point y[n-1].  trace has n-1 values. */
uint32_t orc_virtual_point(const uint32_t *trace, uint32_t log_n);
/* Fibonacci-square trace (prover.rs:32-39): count values. */
void orc_trace_fibsq(uint32_t a0, uint32_t a1, size_t count, uint32_t *out);
/* trace (n-1 values) -> N = n<<log_b coset evaluations f(w*h^i), natural order. */
void orc_lde(const uint32_t *trace, uint32_t log_n, uint32_t log_b, uint32_t *out_evals);
/* Pointwise composition on the coset (prover.rs:101-166 / proof.rs:63-77). */
void orc_compose(const uint32_t *f_eval, uint32_t log_n, uint32_t log_b,
                 const uint32_t alpha_raw[3], uint32_t public_last, uint32_t *cp_out);
/* Evaluation-form fold of layer `round` (m = N>>round values) -> m/2 values. */
void orc_fri_fold_eval(const uint32_t *layer, uint32_t log_n, uint32_t log_b, uint32_t round,
                       uint32_t beta_raw, uint32_t *out);

/* ---- sha2 / merkle.rs -------------------------------------------------- */
void orc_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
/* Merkle::new (merkle.rs:14-51): nodes = (2m-1)*32 bytes, heap order. */
int orc_merkle_build(const uint32_t *vals, size_t m, uint8_t *nodes);
/* Merkle::trace (merkle.rs:54-71): returns path length (log2 m). */
size_t orc_merkle_trace(const uint8_t *nodes, size_t m, size_t leaf, uint8_t *path_out);
/* merkle.rs:42-45: parent of one pair of digests, with the hash selected by orc_set_hash */
void orc_node_hash(const uint8_t *l, const uint8_t *r, uint8_t out[32]);
/* compute_root_from_path (merkle.rs:82-110). */
void orc_compute_root_from_path(uint32_t element, size_t index, const uint8_t *path,
                                size_t path_len, uint8_t out[32]);

/* Merkle hash selector for every function above and below: 0 = SHA-256 (the reference,
 * merkle.rs:1-2), 1 = the build's field-native Poseidon2-style hash (BASELINE.json configs[4];
 * self-defined, no reference counterpart).  The transcript always hashes with SHA-256. */
void orc_set_hash(int kind);
void orc_fieldhash_permute(uint32_t state[16]);
/* orc_merkle_build with the field hash: eight hashes at a time in exact double arithmetic on AVX-512 registers (default
 * on where the CPU has AVX-512F; same digests as the scalar code, which on = 0 selects). */
void orc_set_fieldhash_batch(int on);

/* ---- channel.rs -------------------------------------------------------- */
typedef struct orc_channel {
    uint8_t state[32];
    uint8_t *data;
    size_t len, cap;
} orc_channel;
void orc_channel_new(orc_channel *c);
void orc_channel_free(orc_channel *c);
void orc_channel_commit_bytes(orc_channel *c, const uint8_t *bytes, size_t n); /* channel.rs:19-26 */
uint32_t orc_channel_get_u32(orc_channel *c);                                  /* channel.rs:28-32 */

/* ---- prover.rs / proof.rs ---------------------------------------------- */
typedef struct orc_debug {
    uint32_t *trace;       /* n-1, optional */
    uint32_t *f_eval;      /* N, optional */
    uint32_t *cp_layers;   /* N + N/2 + ... + B values, optional */
    uint8_t *roots;        /* (R+2)*32: f root, cp root, R fri roots; optional */
    uint32_t alpha_raw[3];
    uint32_t beta_raw[32];
    uint32_t free_term;
    uint32_t query_raw;
    uint32_t public_last;  /* a[n-2] */
    uint32_t cp_degree;    /* degree of cp (interpolated), filled in NAIVE mode only */
} orc_debug;

/* generate_proof (prover.rs:9-293) generalised to (log_n, log_b); a0/a1 seed
 * the trace (reference: 1, 3141592).  proof_out receives Channel.data;
 * final_state receives Channel.state.  Returns 0, or <0 on a failed internal
 * check (the reference panics). */
int orc_prove(uint32_t log_n, uint32_t log_b, uint32_t a0, uint32_t a1, int mode,
              uint8_t *proof_out, size_t cap, size_t *proof_len, uint8_t final_state[32],
              orc_debug *dbg);
/* The same prover on a channel that first commits `prefix` (generate_proof takes the caller's Channel, prover.rs:9). */
int orc_prove_prefixed(const uint8_t *prefix, size_t prefix_len, uint32_t log_n, uint32_t log_b, uint32_t a0, uint32_t a1,
                       uint8_t *proof_out, size_t cap, size_t *proof_len, uint8_t final_state[32]);
/* Proof::verify (proof.rs:15-149) generalised; returns 0 if accepted, else the
 * negative number of the first failing check. */
int orc_verify(const uint8_t *data, size_t len, uint32_t log_n, uint32_t log_b,
               uint32_t public_last);
/* Proof::size (proof.rs:151-154): 48 + len on a 64-bit target. */
/* Number of queries of the decommitment (default 1 = the reference, prover.rs:263); see the .c file. */
void orc_set_queries(uint32_t q);
size_t orc_proof_size(size_t data_len);
size_t orc_proof_data_len(uint32_t log_n, uint32_t log_b);

#ifdef __cplusplus
}
#endif
#endif
