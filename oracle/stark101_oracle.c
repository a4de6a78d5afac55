/*
 * stark101_oracle.c -- CPU oracle for the STARK-101 prover hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see stark101_oracle.h).  Plain C restatement of
 * Crocodoctopus/zkstark; every function cites the reference lines it follows
 * (paths relative to /root/reference).  Nothing under zkstark_amd/ uses it.
 *
 * Parity pin: known answers of the reference's own tests and assert_eq!
 * checkpoints (tests/test_oracle_*.py).  Transcript bytes: "parity unpinned"
 * (no golden bytes exist in the reference; bincode 1.x defaults assumed).
 */
#include "stark101_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* field.rs                                                                 */
/* ======================================================================== */

/* field.rs:99-111 Add.  Internal Montgomery form (num-modular) is not
 * observable; only residue() escapes (field.rs:41-43), so plain residues. */
uint32_t orc_add(uint32_t a, uint32_t b, uint32_t p) {
    uint64_t s = (uint64_t)a + b;
    return (uint32_t)(s >= p ? s - p : s);
}
/* field.rs:113-132 Sub */
uint32_t orc_sub(uint32_t a, uint32_t b, uint32_t p) {
    return a >= b ? a - b : (uint32_t)((uint64_t)a + p - b);
}
/* field.rs:134-167 Mul */
uint32_t orc_mul(uint32_t a, uint32_t b, uint32_t p) {
    return (uint32_t)(((uint64_t)a * b) % p);
}
/* field.rs:198-203 Neg */
uint32_t orc_neg(uint32_t a, uint32_t p) { return a ? p - a : 0; }
/* field.rs:26-38 Pow: square-and-multiply per call. */
uint32_t orc_pow(uint32_t a, uint32_t e, uint32_t p) {
    uint32_t r = 1 % p, b = a;
    while (e) {
        if (e & 1) r = orc_mul(r, b, p);
        b = orc_mul(b, b, p);
        e >>= 1;
    }
    return r;
}
/* field.rs:205-210 Inv (p prime: Fermat). */
uint32_t orc_inv(uint32_t a, uint32_t p) { return orc_pow(a, p - 2, p); }
/* field.rs:20-24 From<u32>: MontgomeryInt::new reduces n mod m. */
uint32_t orc_from_u32(uint32_t v, uint32_t p) { return v % p; }
/* field.rs:10-18 From<i32> */
uint32_t orc_from_i32(int32_t v, uint32_t p) {
    if (v < 0) return orc_neg((uint32_t)(-(int64_t)v) % p, p);
    return (uint32_t)v % p;
}
/* field.rs:165-177 Div: MontgomeryInt division = product with the inverse; a zero divisor panics there (returns 0 here). */
uint32_t orc_div(uint32_t a, uint32_t b, uint32_t p) { return b % p ? orc_mul(a % p, orc_inv(b % p, p), p) : 0; }
/* field.rs:89-94 Rem<u32>: convert(residue % rhs); rhs = 0 panics there (returns 0 here). */
uint32_t orc_rem(uint32_t a, uint32_t rhs, uint32_t p) { return rhs ? (a % p) % rhs % p : 0; }
/* field.rs:45-49 order(): brute force. */
uint32_t orc_order(uint32_t a, uint32_t p) {
    uint32_t x = a % p;
    for (uint32_t it = 1;; ++it) {
        if (x == 1) return it;
        x = orc_mul(x, a, p);
        if (it == 0xffffffffu) return 0;
    }
}
/* field.rs:52-86 generator(): smallest x >= 2 with x^((p-1)/f) != 1 for all
 * unique prime factors f of p-1. */
uint32_t orc_generator(uint32_t p) {
    uint32_t factors[32];
    int nf = 0;
    uint32_t q = p - 1, it = 2;
    while (q != 1) {                      /* field.rs:57-65 */
        if (q % it == 0) factors[nf++] = it;
        while (q % it == 0) q /= it;
        it += 1;
    }
    uint32_t exps[32];
    for (int i = 0; i < nf; ++i)          /* field.rs:68-73: (p-1)*inv(f) == (p-1)/f */
        exps[i] = orc_mul(p - 1, orc_inv(factors[i], p), p);
    for (uint32_t x = 2; x < p; ++x) {    /* field.rs:76-84 */
        int ok = 1;
        for (int i = 0; i < nf; ++i)
            if (orc_pow(x, exps[i], p) == 1) { ok = 0; break; }
        if (ok) return x;
    }
    return 0;
}

/* Fast fixed-modulus helpers (compiler turns % ORC_P into multiply/shift). */
static inline uint32_t fadd(uint32_t a, uint32_t b) {
    uint64_t s = (uint64_t)a + b;
    return (uint32_t)(s >= ORC_P ? s - ORC_P : s);
}
static inline uint32_t fsub(uint32_t a, uint32_t b) {
    return a >= b ? a - b : (uint32_t)((uint64_t)a + ORC_P - b);
}
static inline uint32_t fmul(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * b) % ORC_P);
}
static uint32_t fpow(uint32_t a, uint64_t e) {
    uint32_t r = 1, b = a;
    while (e) {
        if (e & 1) r = fmul(r, b);
        b = fmul(b, b);
        e >>= 1;
    }
    return r;
}
static inline uint32_t finv(uint32_t a) { return fpow(a, ORC_P - 2); }

/* Montgomery's batch inversion trick, in place; all inputs non-zero. */
static void batch_inv(uint32_t *v, size_t n, uint32_t *scratch) {
    if (!n) return;
    uint32_t acc = 1;
    for (size_t i = 0; i < n; ++i) { scratch[i] = acc; acc = fmul(acc, v[i]); }
    uint32_t inv = finv(acc);
    for (size_t i = n; i-- > 0;) {
        uint32_t t = fmul(inv, scratch[i]);
        inv = fmul(inv, v[i]);
        v[i] = t;
    }
}

/* ======================================================================== */
/* polynomial.rs -- literal (naive) restatement                             */
/* ======================================================================== */

typedef struct { uint32_t *c; size_t len; } poly;  /* low degree first (polynomial.rs:31) */

static poly poly_new(size_t len) {
    poly r; r.len = len;
    r.c = (uint32_t *)calloc(len ? len : 1, sizeof(uint32_t));
    return r;
}
static void poly_free(poly *a) { free(a->c); a->c = NULL; a->len = 0; }
static poly poly_clone(const poly *a) {
    poly r = poly_new(a->len);
    memcpy(r.c, a->c, a->len * sizeof(uint32_t));
    return r;
}
/* polynomial.rs:15-28 reduce(): strip zero leading coefficients. */
static void poly_reduce(poly *a) {
    while (a->len && a->c[a->len - 1] == 0) a->len--;
}
/* polynomial.rs:97-128 Add (with the None shortcuts). */
static poly poly_add(const poly *a, const poly *b, uint32_t p) {
    if (!a->len && !b->len) return poly_new(0);
    if (!a->len) return poly_clone(b);
    if (!b->len) return poly_clone(a);
    poly r = poly_new(a->len > b->len ? a->len : b->len);
    for (size_t i = 0; i < a->len; ++i) r.c[i] = orc_add(r.c[i], a->c[i], p);
    for (size_t i = 0; i < b->len; ++i) r.c[i] = orc_add(r.c[i], b->c[i], p);
    poly_reduce(&r);
    return r;
}
/* polynomial.rs:163-195 Sub.  Quirk kept: (None, Some) returns rhs un-negated. */
static poly poly_sub(const poly *a, const poly *b, uint32_t p) {
    if (!a->len && !b->len) return poly_new(0);
    if (!a->len) return poly_clone(b);
    if (!b->len) return poly_clone(a);
    poly r = poly_new(a->len > b->len ? a->len : b->len);
    for (size_t i = 0; i < a->len; ++i) r.c[i] = orc_add(r.c[i], a->c[i], p);
    for (size_t i = 0; i < b->len; ++i) r.c[i] = orc_sub(r.c[i], b->c[i], p);
    poly_reduce(&r);
    return r;
}
/* polynomial.rs:230-260 Mul: full schoolbook double loop, no reduce. */
static poly poly_mul(const poly *a, const poly *b, uint32_t p) {
    if (!a->len || !b->len) return poly_new(0);
    poly r = poly_new(a->len + b->len - 1);
    for (size_t i = 0; i < a->len; ++i) {
        uint64_t ai = a->c[i];
        for (size_t j = 0; j < b->len; ++j)
            r.c[i + j] = orc_add(r.c[i + j], (uint32_t)((ai * b->c[j]) % p), p);
    }
    return r;
}
/* polynomial.rs:6-13 x(t, e): t * x^e. */
static poly poly_monomial(uint32_t t, size_t e) {
    poly r = poly_new(e + 1);
    r.c[e] = t;
    return r;
}
/* Polynomial::from([1, -root]) = x - root (polynomial.rs:38-40 reverses). */
static poly poly_linear(uint32_t root, uint32_t p) {
    poly r = poly_new(2);
    r.c[0] = orc_neg(root, p);
    r.c[1] = 1 % p;
    return r;
}
/* polynomial.rs:49-56 solve(): sum coef_d * x^d with a fresh pow per term. */
static uint32_t poly_solve(const poly *a, uint32_t x, uint32_t p) {
    uint32_t acc = 0;
    for (size_t d = 0; d < a->len; ++d)
        acc = orc_add(acc, orc_mul(a->c[d], orc_pow(x, (uint32_t)d, p), p), p);
    return acc;
}
uint32_t orc_poly_solve_naive(const uint32_t *coef, size_t len, uint32_t x, uint32_t p) {
    poly a; a.c = (uint32_t *)coef; a.len = len;
    return poly_solve(&a, x, p);
}
/* polynomial.rs:74-79 apply_const(): coef_d *= t^d. */
static void poly_apply_const(poly *a, uint32_t t, uint32_t p) {
    for (size_t d = 0; d < a->len; ++d)
        a->c[d] = orc_mul(a->c[d], orc_pow(t, (uint32_t)d, p), p);
}
/* polynomial.rs:290-300 MulAssign<T>. */
static void poly_scale(poly *a, uint32_t s, uint32_t p) {
    for (size_t d = 0; d < a->len; ++d) a->c[d] = orc_mul(a->c[d], s, p);
}

/* polynomial.rs:307-334 div(): the recursion unrolled into a loop, same
 * arithmetic per step (quotient monomial times the whole divisor through the
 * schoolbook Mul, zeros included). */
static int poly_div(const poly *num, const poly *den, poly *quot, poly *rem, uint32_t p) {
    poly lhs = poly_clone(num);
    size_t rdeg = den->len ? den->len - 1 : 0;
    size_t ldeg0 = lhs.len ? lhs.len - 1 : 0;
    poly q = poly_new(ldeg0 >= rdeg ? ldeg0 - rdeg + 1 : 0);
    size_t qlen = 0;
    for (;;) {
        size_t ldeg = lhs.len ? lhs.len - 1 : 0;
        if (ldeg < rdeg) break;
        if (!lhs.len || !den->len) { poly_free(&lhs); poly_free(&q); return -1; } /* ref panics */
        uint32_t lead = orc_mul(lhs.c[lhs.len - 1], orc_inv(den->c[den->len - 1], p), p);
        size_t diff = ldeg - rdeg;
        poly d = poly_monomial(lead, diff);
        poly prod = poly_mul(&d, den, p);
        poly r = poly_sub(&lhs, &prod, p);
        if (diff + 1 > qlen) qlen = diff + 1;
        q.c[diff] = orc_add(q.c[diff], lead, p);
        poly_free(&d); poly_free(&prod); poly_free(&lhs);
        lhs = r;
    }
    q.len = qlen;
    poly_reduce(&q);
    *quot = q;
    *rem = lhs;
    return 0;
}
size_t orc_poly_div(const uint32_t *num, size_t num_len, const uint32_t *den, size_t den_len,
                    uint32_t *quot_out, uint32_t *rem_out, size_t *rem_len, uint32_t p) {
    poly a, b, q, r;
    a.c = (uint32_t *)num; a.len = num_len;
    b.c = (uint32_t *)den; b.len = den_len;
    if (poly_div(&a, &b, &q, &r, p)) { *rem_len = (size_t)-1; return 0; }
    memcpy(quot_out, q.c, q.len * sizeof(uint32_t));
    memcpy(rem_out, r.c, r.len * sizeof(uint32_t));
    *rem_len = r.len;
    size_t ql = q.len;
    poly_free(&q); poly_free(&r);
    return ql;
}
/* Same routine over i32 with truncating integer division (div_test). */
size_t orc_poly_div_i32(const int32_t *num, size_t num_len, const int32_t *den, size_t den_len,
                        int32_t *quot_out, int32_t *rem_out, size_t *rem_len) {
    int32_t *lhs = (int32_t *)malloc((num_len + 1) * sizeof(int32_t));
    memcpy(lhs, num, num_len * sizeof(int32_t));
    size_t llen = num_len, qlen = 0;
    size_t rdeg = den_len ? den_len - 1 : 0;
    memset(quot_out, 0, (num_len ? num_len : 1) * sizeof(int32_t));
    for (;;) {
        size_t ldeg = llen ? llen - 1 : 0;
        if (ldeg < rdeg || !llen) break;
        int32_t lead = lhs[llen - 1] / den[den_len - 1];
        size_t diff = ldeg - rdeg;
        for (size_t j = 0; j < den_len; ++j) lhs[diff + j] -= lead * den[j];
        quot_out[diff] += lead;
        if (diff + 1 > qlen) qlen = diff + 1;
        size_t before = llen;
        while (llen && lhs[llen - 1] == 0) llen--;
        if (llen == before) break; /* inexact integer lead: stop like a non-terminating ref */
    }
    memcpy(rem_out, lhs, llen * sizeof(int32_t));
    *rem_len = llen;
    free(lhs);
    while (qlen && quot_out[qlen - 1] == 0) qlen--;
    return qlen;
}

/* polynomial.rs:337-383 lagrange(): ll/lr prefix and suffix products, combine
 * by schoolbook Mul, normalise by 1/basis(x_i), scale by y_i, sum. */
static poly lagrange_naive(const uint32_t *xs, const uint32_t *ys, size_t npts, uint32_t p) {
    poly *ll = (poly *)malloc(npts * sizeof(poly));
    poly *lr = (poly *)malloc(npts * sizeof(poly));
    ll[0] = poly_new(1); ll[0].c[0] = 1 % p;
    for (size_t i = 1; i < npts; ++i) {               /* :354-356 */
        poly lin = poly_linear(xs[i - 1], p);
        ll[i] = poly_mul(&ll[i - 1], &lin, p);
        poly_free(&lin);
    }
    lr[npts - 1] = poly_new(1); lr[npts - 1].c[0] = 1 % p;
    for (size_t i = npts - 1; i-- > 0;) {             /* :360-362 */
        poly lin = poly_linear(xs[i + 1], p);
        lr[i] = poly_mul(&lr[i + 1], &lin, p);
        poly_free(&lin);
    }
    poly acc = poly_new(0);
    for (size_t i = 0; i < npts; ++i) {
        poly basis = poly_mul(&ll[i], &lr[i], p);     /* :365 */
        poly_scale(&basis, orc_inv(poly_solve(&basis, xs[i], p), p), p);  /* :369-371 */
        poly_scale(&basis, ys[i], p);                 /* :377-380 */
        if (i == 0) { poly_free(&acc); acc = basis; } /* reduce(): first element as is */
        else {
            poly s = poly_add(&acc, &basis, p);       /* :381 */
            poly_free(&acc); poly_free(&basis);
            acc = s;
        }
        poly_free(&ll[i]); poly_free(&lr[i]);
    }
    free(ll); free(lr);
    return acc;
}
void orc_lagrange_naive(const uint32_t *xs, const uint32_t *ys, size_t npts,
                        uint32_t *coef_out, uint32_t p) {
    poly r = lagrange_naive(xs, ys, npts, p);
    memset(coef_out, 0, npts * sizeof(uint32_t));
    memcpy(coef_out, r.c, r.len * sizeof(uint32_t));
    poly_free(&r);
}

/* polynomial.rs:385-400 fri(): out[i] = c[2i] + b*c[2i+1], len/2 (integer). */
void orc_fri_coef_fold(const uint32_t *coef, size_t len, uint32_t beta, uint32_t *out, uint32_t p) {
    for (size_t i = 0; i < len / 2; ++i)
        out[i] = orc_add(coef[2 * i], orc_mul(beta, coef[2 * i + 1], p), p);
}

/* ======================================================================== */
/* Size-generic transforms                           */
/* ======================================================================== */

void orc_set_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

static void bit_reverse(uint32_t *a, uint32_t log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) {
        size_t j = 0, x = i;
        for (uint32_t b = 0; b < log_n; ++b) { j = (j << 1) | (x & 1); x >>= 1; }
        if (j > i) { uint32_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

/* Textbook iterative radix-2 DIT: X[k] = sum_j x[j] root^(jk). */
void orc_ntt(uint32_t *a, uint32_t log_n, uint32_t root) {
    size_t n = (size_t)1 << log_n;
    if (n == 1) return;
    uint32_t *tw = (uint32_t *)malloc((n / 2) * sizeof(uint32_t));
    tw[0] = 1;
    for (size_t i = 1; i < n / 2; ++i) tw[i] = fmul(tw[i - 1], root);
    bit_reverse(a, log_n);
    for (uint32_t s = 1; s <= log_n; ++s) {
        size_t half = (size_t)1 << (s - 1), step = n >> s;
#pragma omp parallel for schedule(static) if (n >= (1u << 16))
        for (size_t blk = 0; blk < n; blk += 2 * half) {
            for (size_t j = 0; j < half; ++j) {
                uint32_t u = a[blk + j];
                uint32_t v = fmul(a[blk + j + half], tw[j * step]);
                a[blk + j] = fadd(u, v);
                a[blk + j + half] = fsub(u, v);
            }
        }
    }
    free(tw);
}
void orc_intt(uint32_t *a, uint32_t log_n, uint32_t root) {
    size_t n = (size_t)1 << log_n;
    orc_ntt(a, log_n, finv(root));
    uint32_t ninv = finv((uint32_t)(n % ORC_P));
    for (size_t i = 0; i < n; ++i) a[i] = fmul(a[i], ninv);
}

/* prover.rs:32-39 */
void orc_trace_fibsq(uint32_t a0, uint32_t a1, size_t count, uint32_t *out) {
    if (count > 0) out[0] = a0 % ORC_P;
    if (count > 1) out[1] = a1 % ORC_P;
    for (size_t i = 2; i < count; ++i)
        out[i] = fadd(fmul(out[i - 2], out[i - 2]), fmul(out[i - 1], out[i - 1]));
}

/* generators: w = 5 (prover.rs:44-45), g = w^((P-1)/n), h = w^((P-1)/N)
 * (prover.rs:48-49 with 3145728 = (P-1)/1024 and 393216 = (P-1)/8192). */
static uint32_t gen_w(void) { return 5; }
static uint32_t gen_of_order_log(uint32_t log_order) {
    return fpow(gen_w(), (uint64_t)(ORC_P - 1) >> log_order);
}

/* the reference interpolates n-1 points (prover.rs:60 zip
 * truncates), so the degree-(n-1) coefficient of the size-n interpolant must
 * vanish: y[n-1] = -g * sum_{i<n-1} a_i g^i. */
uint32_t orc_virtual_point(const uint32_t *trace, uint32_t log_n) {
    size_t n = (size_t)1 << log_n;
    uint32_t g = gen_of_order_log(log_n), gi = 1, acc = 0;
    for (size_t i = 0; i + 1 < n; ++i) {
        acc = fadd(acc, fmul(trace[i], gi));
        gi = fmul(gi, g);
    }
    return fsub(0, fmul(g, acc));
}

/* lagrange (polynomial.rs:337) + solve over w*h^i (prover.rs:69-70)
 * == iNTT_g, scale coef k by w^k, zero-pad, NTT_h. */
void orc_lde(const uint32_t *trace, uint32_t log_n, uint32_t log_b, uint32_t *out) {
    size_t n = (size_t)1 << log_n, N = n << log_b;
    uint32_t g = gen_of_order_log(log_n), h = gen_of_order_log(log_n + log_b);
    memset(out, 0, N * sizeof(uint32_t));
    memcpy(out, trace, (n - 1) * sizeof(uint32_t));
    out[n - 1] = orc_virtual_point(trace, log_n);
    orc_intt(out, log_n, g);
    uint32_t wk = 1;
    for (size_t k = 0; k < n; ++k) { out[k] = fmul(out[k], wk); wk = fmul(wk, gen_w()); }
    orc_ntt(out, log_n + log_b, h);
}

/* prover.rs:101-166 evaluated pointwise on the coset, which is what the
 * verifier recomputes at proof.rs:63-77:
 *   p0 = (f(x)-1)/(x-g^0)                         (prover.rs:101-103; a[0] = trace[0])
 *   p1 = (f(x)-a[n-2])/(x-g^(n-2))                (prover.rs:111-113)
 *   p2 = (f(g^2 x)-f(gx)^2-f(x)^2)
 *        * (x-g^(n-3))(x-g^(n-2))(x-g^(n-1)) / (x^n-1)   (prover.rs:134-145)
 *   cp = a0*p0 + a1*p1 + a2*p2                    (prover.rs:163-166)
 * with f(g x_i) = f_eval[i+B], f(g^2 x_i) = f_eval[i+2B] (indices mod N). */
void orc_compose(const uint32_t *f, uint32_t log_n, uint32_t log_b, const uint32_t alpha_raw[3],
                 uint32_t public_last, uint32_t *cp) {
    size_t n = (size_t)1 << log_n, B = (size_t)1 << log_b, N = n << log_b;
    uint32_t w = gen_w(), g = gen_of_order_log(log_n), h = gen_of_order_log(log_n + log_b);
    uint32_t a0 = alpha_raw[0] % ORC_P, a1 = alpha_raw[1] % ORC_P, a2 = alpha_raw[2] % ORC_P;
    uint32_t g_n1 = finv(g), g_n2 = fmul(g_n1, g_n1), g_n3 = fmul(g_n2, g_n1);
    uint32_t first = 1; /* a[0]: prover.rs:33, verifier literal proof.rs:69 */
    /* x^n takes B values: (w h^i)^n = w^n (h^n)^(i mod B) */
    uint32_t *zinv = (uint32_t *)malloc(B * sizeof(uint32_t));
    uint32_t wn = fpow(w, n), hn = fpow(h, n), t = wn;
    for (size_t r = 0; r < B; ++r) { zinv[r] = finv(fsub(t, 1)); t = fmul(t, hn); }
    const size_t CH = 4096;
#pragma omp parallel
    {
        uint32_t *den = (uint32_t *)malloc(2 * CH * sizeof(uint32_t));
        uint32_t *scr = (uint32_t *)malloc(2 * CH * sizeof(uint32_t));
        uint32_t *xs = (uint32_t *)malloc(CH * sizeof(uint32_t));
#pragma omp for schedule(static)
        for (size_t c0 = 0; c0 < N; c0 += CH) {
            size_t cnt = N - c0 < CH ? N - c0 : CH;
            uint32_t x = fmul(w, fpow(h, c0));
            for (size_t k = 0; k < cnt; ++k) {
                xs[k] = x;
                den[2 * k] = fsub(x, 1);
                den[2 * k + 1] = fsub(x, g_n2);
                x = fmul(x, h);
            }
            batch_inv(den, 2 * cnt, scr);
            for (size_t k = 0; k < cnt; ++k) {
                size_t i = c0 + k;
                uint32_t f0 = f[i], f1 = f[(i + B) & (N - 1)], f2 = f[(i + 2 * B) & (N - 1)];
                uint32_t xx = xs[k];
                uint32_t p0 = fmul(fsub(f0, first), den[2 * k]);
                uint32_t p1 = fmul(fsub(f0, public_last), den[2 * k + 1]);
                uint32_t num = fsub(fsub(f2, fmul(f1, f1)), fmul(f0, f0));
                uint32_t v3 = fmul(fmul(fsub(xx, g_n3), fsub(xx, g_n2)), fsub(xx, g_n1));
                uint32_t p2 = fmul(fmul(num, v3), zinv[i & (B - 1)]);
                cp[i] = fadd(fadd(fmul(a0, p0), fmul(a1, p1)), fmul(a2, p2));
            }
        }
        free(den); free(scr); free(xs);
    }
    free(zinv);
}

/* polynomial.rs:385-400 + prover.rs:204-211 in evaluation form;
 * the identity is the one pinned by fri_test (polynomial.rs:418-425) and used
 * by the verifier (proof.rs:110-113):
 *   next[i] = (e[i]+e[i+m/2])/2 + beta*(e[i]-e[i+m/2])/(2 x_i),
 *   x_i = (w h^i)^(2^round),  m = N >> round. */
void orc_fri_fold_eval(const uint32_t *e, uint32_t log_n, uint32_t log_b, uint32_t round,
                       uint32_t beta_raw, uint32_t *out) {
    size_t N = (size_t)1 << (log_n + log_b), m = N >> round, half = m / 2;
    uint32_t h = gen_of_order_log(log_n + log_b);
    uint32_t beta = beta_raw % ORC_P, inv2 = finv(2);
    uint32_t x0inv = finv(fpow(gen_w(), (uint64_t)1 << round));
    uint32_t hrinv = finv(fpow(h, (uint64_t)1 << round));
    const size_t CH = 8192;
#pragma omp parallel for schedule(static) if (half >= 65536)
    for (size_t c0 = 0; c0 < half; c0 += CH) {
        size_t cnt = half - c0 < CH ? half - c0 : CH;
        uint32_t xinv = fmul(x0inv, fpow(hrinv, c0));
        for (size_t k = 0; k < cnt; ++k) {
            size_t i = c0 + k;
            uint32_t s = fmul(fadd(e[i], e[i + half]), inv2);
            uint32_t d = fmul(fmul(fsub(e[i], e[i + half]), inv2), xinv);
            out[i] = fadd(s, fmul(beta, d));
            xinv = fmul(xinv, hrinv);
        }
    }
}

/* ======================================================================== */
/* SHA-256 (FIPS 180-4; the sha2 crate, merkle.rs:1-2, channel.rs:4)        */
/* ======================================================================== */

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#define ROTR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void sha256_compress(uint32_t st[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) |
               ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = ROTR(w[i - 15], 7) ^ ROTR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROTR(w[i - 2], 17) ^ ROTR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t t1 = h + (ROTR(e, 6) ^ ROTR(e, 11) ^ ROTR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ROTR(a, 2) ^ ROTR(a, 13) ^ ROTR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

typedef struct { uint32_t st[8]; uint8_t buf[64]; size_t fill; uint64_t total; } sha_ctx;
static void sha_init(sha_ctx *c) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                   0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(c->st, iv, sizeof iv);
    c->fill = 0; c->total = 0;
}
static void sha_update(sha_ctx *c, const uint8_t *m, size_t n) {
    c->total += n;
    while (n) {
        size_t take = 64 - c->fill < n ? 64 - c->fill : n;
        memcpy(c->buf + c->fill, m, take);
        c->fill += take; m += take; n -= take;
        if (c->fill == 64) { sha256_compress(c->st, c->buf); c->fill = 0; }
    }
}
static void sha_final(sha_ctx *c, uint8_t out[32]) {
    uint64_t bits = c->total * 8;
    uint8_t pad = 0x80;
    sha_update(c, &pad, 1);
    uint8_t z = 0;
    while (c->fill != 56) sha_update(c, &z, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha_update(c, len, 8);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(c->st[i] >> 24); out[4 * i + 1] = (uint8_t)(c->st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(c->st[i] >> 8); out[4 * i + 3] = (uint8_t)c->st[i];
    }
}
void orc_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    sha_ctx c; sha_init(&c); sha_update(&c, msg, len); sha_final(&c, out);
}

/* ======================================================================== */
/* Field-native Merkle hash (BASELINE.json configs[4])  */
/* ======================================================================== */
/* The reference has only SHA-256 (merkle.rs:1-2).  This Poseidon2-style permutation over GF(P)
 * is the build's own definition ("parity: self-defined"); the spec is DESIGN.md section 7 and
 * this code.  Width 16, S-box x^5 (gcd(5, P-1) = 1), 8 full + 22 partial rounds, external layer
 * circ(2 M4, M4, M4, M4), internal layer J + diag(d).  A performance stand-in, not a vetted
 * instance.  Digest = first 8 state words of perm(state) + state. */
enum { ORC_HASH_SHA256 = 0, ORC_HASH_FIELD = 1 };
static int g_hash_kind = ORC_HASH_SHA256;
void orc_set_hash(int kind) { g_hash_kind = kind; }

#define FH_T 16
#define FH_RF 8
#define FH_RP 22
static uint32_t fh_rc_full[FH_RF][FH_T], fh_rc_part[FH_RP];
static int fh_ready = 0;
static const uint32_t fh_diag[FH_T] = {ORC_P - 2, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384};

static void fh_init(void) {
    if (fh_ready) return;
    const char *tag = "zkstark_amd.fieldhash.v1";
    size_t tl = strlen(tag);
    for (uint32_t k = 0; k < FH_RF * FH_T + FH_RP; ++k) {
        uint8_t msg[64], dig[32];
        memcpy(msg, tag, tl);
        msg[tl] = (uint8_t)k; msg[tl + 1] = (uint8_t)(k >> 8); msg[tl + 2] = (uint8_t)(k >> 16); msg[tl + 3] = (uint8_t)(k >> 24);
        orc_sha256(msg, tl + 4, dig);
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | dig[i];
        uint32_t c = (uint32_t)(v % ORC_P);
        if (k < FH_RF * FH_T) fh_rc_full[k / FH_T][k % FH_T] = c; else fh_rc_part[k - FH_RF * FH_T] = c;
    }
    fh_ready = 1;
}
static inline uint32_t fh_sbox(uint32_t x) { uint32_t x2 = fmul(x, x); return fmul(fmul(x2, x2), x); }
/* 4x4 block: the linear map defined by this sequence of additions and doublings */
static void fh_m4(uint32_t *x) {
    uint32_t t0 = fadd(x[0], x[1]), t1 = fadd(x[2], x[3]);
    uint32_t t2 = fadd(fadd(x[1], x[1]), t1), t3 = fadd(fadd(x[3], x[3]), t0);
    uint32_t t1_4 = fadd(t1, t1); t1_4 = fadd(t1_4, t1_4);
    uint32_t t0_4 = fadd(t0, t0); t0_4 = fadd(t0_4, t0_4);
    uint32_t t4 = fadd(t1_4, t3), t5 = fadd(t0_4, t2);
    uint32_t t6 = fadd(t3, t5), t7 = fadd(t2, t4);
    x[0] = t6; x[1] = t5; x[2] = t7; x[3] = t4;
}
static void fh_external(uint32_t *s) {
    for (int b = 0; b < 4; ++b) fh_m4(s + 4 * b);
    uint32_t col[4];
    for (int j = 0; j < 4; ++j) col[j] = fadd(fadd(s[j], s[4 + j]), fadd(s[8 + j], s[12 + j]));
    for (int i = 0; i < FH_T; ++i) s[i] = fadd(s[i], col[i & 3]);
}
static void fh_internal(uint32_t *s) {
    uint32_t sum = 0;
    for (int i = 0; i < FH_T; ++i) sum = fadd(sum, s[i]);
    for (int i = 0; i < FH_T; ++i) s[i] = fadd(fmul(s[i], fh_diag[i]), sum);
}
void orc_fieldhash_permute(uint32_t s[16]) {
    fh_init();
    fh_external(s);
    for (int r = 0; r < FH_RF / 2; ++r) {
        for (int i = 0; i < FH_T; ++i) s[i] = fh_sbox(fadd(s[i], fh_rc_full[r][i]));
        fh_external(s);
    }
    for (int r = 0; r < FH_RP; ++r) {
        s[0] = fh_sbox(fadd(s[0], fh_rc_part[r]));
        fh_internal(s);
    }
    for (int r = FH_RF / 2; r < FH_RF; ++r) {
        for (int i = 0; i < FH_T; ++i) s[i] = fh_sbox(fadd(s[i], fh_rc_full[r][i]));
        fh_external(s);
    }
}
static void fh_digest(const uint32_t in[16], uint8_t out[32]) {
    uint32_t s[16];
    memcpy(s, in, sizeof s);
    orc_fieldhash_permute(s);
    for (int i = 0; i < 8; ++i) {
        uint32_t v = fadd(s[i], in[i]);
        out[4 * i] = (uint8_t)(v >> 24); out[4 * i + 1] = (uint8_t)(v >> 16); out[4 * i + 2] = (uint8_t)(v >> 8); out[4 * i + 3] = (uint8_t)v;
    }
}
static uint32_t fh_word(const uint8_t *p) { return (((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]) % ORC_P; }

/* ---- the same permutation, EIGHT hashes at a time, in exact double-precision arithmetic ---------------------------
 * Test infrastructure for BASELINE.json configs[4] at its stated size (domain 2^24 = 10^8 hashes per proof): the scalar
 * code above manages ~0.3 M hashes/s per core.  This form is still a plain-residue restatement and shares nothing with
 * the GPU code (which is 32-bit Montgomery): every state element is an integer held exactly in a double (< 2^53),
 * linear layers accumulate without reduction and reduce once, and a product a*b mod P is formed from the exact
 * split a*b = h + l (h = fl(a*b), l = fma(a, b, -h)) and the quotient estimate q = rint(h / P):
 *     r = fma(-q, P, h) + l   is exactly a*b - q*P,  |r| < P.
 * The eight lanes of an AVX-512 register are eight independent hashes (x86-64 with AVX-512F only; any other CPU keeps
 * the scalar code).  tests/test_oracle_reference.py pins it against the scalar code above on random values and on
 * whole trees; orc_set_fieldhash_batch(0) switches it off. */
#define FHW 8
static int g_fh_batch = 1;
void orc_set_fieldhash_batch(int on) { g_fh_batch = on != 0; }
#if defined(__x86_64__)
#include <immintrin.h>
#define FH_HAVE_BATCH 1
#define FH_AVX512 __attribute__((target("avx512f"), always_inline)) static inline
typedef __m512d fhv;                                   /* one state element of eight hashes */
/* x (an integer, |x| < 2^51) -> x mod P in [0, P) */
FH_AVX512 fhv fhd_reduce(fhv x, fhv vp, fhv vinvp) {
    fhv q = _mm512_roundscale_pd(_mm512_mul_pd(x, vinvp), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    fhv r = _mm512_fnmadd_pd(q, vp, x);                /* exact: |x - q P| <= P/2 + 1 */
    return _mm512_mask_add_pd(r, _mm512_cmp_pd_mask(r, _mm512_setzero_pd(), _CMP_LT_OQ), r, vp);
}
/* a, b in [0, P) -> a*b mod P in [0, P) */
FH_AVX512 fhv fhd_mul(fhv a, fhv b, fhv vp, fhv vinvp) {
    fhv h = _mm512_mul_pd(a, b);
    fhv l = _mm512_fmsub_pd(a, b, h);                  /* a*b = h + l exactly */
    fhv q = _mm512_roundscale_pd(_mm512_mul_pd(h, vinvp), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    fhv r = _mm512_add_pd(_mm512_fnmadd_pd(q, vp, h), l);   /* exactly a*b - q*P, |.| < P */
    return _mm512_mask_add_pd(r, _mm512_cmp_pd_mask(r, _mm512_setzero_pd(), _CMP_LT_OQ), r, vp);
}
/* (x + c)^5, x and c in [0, P) */
FH_AVX512 fhv fhd_sbox(fhv x, fhv c, fhv vp, fhv vinvp) {
    x = _mm512_add_pd(x, c);
    x = _mm512_mask_sub_pd(x, _mm512_cmp_pd_mask(x, vp, _CMP_GE_OQ), x, vp);
    fhv x2 = fhd_mul(x, x, vp, vinvp);
    return fhd_mul(fhd_mul(x2, x2, vp, vinvp), x, vp, vinvp);
}
/* fh_m4 + the column sums of fh_external, unreduced (coefficients sum to < 2^6: values < 2^38), then one reduction */
FH_AVX512 void fhd_external(fhv s[FH_T], fhv vp, fhv vinvp) {
    const fhv two = _mm512_set1_pd(2.0), four = _mm512_set1_pd(4.0);
    for (int b = 0; b < 4; ++b) {
        fhv x0 = s[4 * b], x1 = s[4 * b + 1], x2 = s[4 * b + 2], x3 = s[4 * b + 3];
        fhv t0 = _mm512_add_pd(x0, x1), t1 = _mm512_add_pd(x2, x3);
        fhv t2 = _mm512_fmadd_pd(two, x1, t1), t3 = _mm512_fmadd_pd(two, x3, t0);     /* small integers: exact */
        fhv t4 = _mm512_fmadd_pd(four, t1, t3), t5 = _mm512_fmadd_pd(four, t0, t2);
        s[4 * b] = _mm512_add_pd(t3, t5); s[4 * b + 1] = t5; s[4 * b + 2] = _mm512_add_pd(t2, t4); s[4 * b + 3] = t4;
    }
    for (int j = 0; j < 4; ++j) {
        fhv col = _mm512_add_pd(_mm512_add_pd(s[j], s[4 + j]), _mm512_add_pd(s[8 + j], s[12 + j]));
        for (int b = 0; b < 4; ++b) s[4 * b + j] = fhd_reduce(_mm512_add_pd(s[4 * b + j], col), vp, vinvp);
    }
}
FH_AVX512 void fhd_internal(fhv s[FH_T], fhv vp, fhv vinvp) {
    fhv sum = s[0];
    for (int i = 1; i < FH_T; ++i) sum = _mm512_add_pd(sum, s[i]);                       /* < 2^36 */
    s[0] = fhd_reduce(_mm512_fnmadd_pd(_mm512_set1_pd(2.0), s[0], sum), vp, vinvp);      /* diag[0] = P - 2 = -2 */
    for (int i = 1; i < FH_T; ++i) s[i] = fhd_reduce(_mm512_fmadd_pd(s[i], _mm512_set1_pd((double)(1u << (i - 1))), sum), vp, vinvp);   /* diag[i] = 2^(i-1): < 2^47 */
}
/* orc_fieldhash_permute on eight states at once: st[i * 8 + h] = element i of hash h */
__attribute__((target("avx512f")))
static void fhd_permute(double *st, const double rcf[FH_RF][FH_T], const double rcp[FH_RP]) {
    const fhv vp = _mm512_set1_pd(3221225473.0), vinvp = _mm512_set1_pd(1.0 / 3221225473.0);
    fhv s[FH_T];
    for (int i = 0; i < FH_T; ++i) s[i] = _mm512_loadu_pd(st + 8 * i);
    fhd_external(s, vp, vinvp);
    for (int r = 0; r < FH_RF / 2; ++r) {
        for (int i = 0; i < FH_T; ++i) s[i] = fhd_sbox(s[i], _mm512_set1_pd(rcf[r][i]), vp, vinvp);
        fhd_external(s, vp, vinvp);
    }
    for (int r = 0; r < FH_RP; ++r) {
        s[0] = fhd_sbox(s[0], _mm512_set1_pd(rcp[r]), vp, vinvp);
        fhd_internal(s, vp, vinvp);
    }
    for (int r = FH_RF / 2; r < FH_RF; ++r) {
        for (int i = 0; i < FH_T; ++i) s[i] = fhd_sbox(s[i], _mm512_set1_pd(rcf[r][i]), vp, vinvp);
        fhd_external(s, vp, vinvp);
    }
    for (int i = 0; i < FH_T; ++i) _mm512_storeu_pd(st + 8 * i, s[i]);
}
static int fh_batch_usable(void) { return g_fh_batch && __builtin_cpu_supports("avx512f"); }
#else
#define FH_HAVE_BATCH 0
static void fhd_permute(double *st, const double rcf[FH_RF][FH_T], const double rcp[FH_RP]) { (void)st; (void)rcf; (void)rcp; }
static int fh_batch_usable(void) { return 0; }
#endif
/* digests of FHW states in[h][16] (canonical residues) -> out + 32 h */
static void fh_digest_batch(uint32_t in[FHW][FH_T], uint8_t *out) {
    static double rcf[FH_RF][FH_T], rcp[FH_RP];
    static int ready = 0;
    if (!ready) {
#pragma omp critical(fh_batch_init)
        if (!ready) {
            fh_init();
            for (int r = 0; r < FH_RF; ++r) for (int i = 0; i < FH_T; ++i) rcf[r][i] = (double)fh_rc_full[r][i];
            for (int r = 0; r < FH_RP; ++r) rcp[r] = (double)fh_rc_part[r];
            __atomic_store_n(&ready, 1, __ATOMIC_RELEASE);
        }
    }
    double s[FH_T * FHW];
    for (int h = 0; h < FHW; ++h) for (int i = 0; i < FH_T; ++i) s[i * FHW + h] = (double)in[h][i];
    fhd_permute(s, rcf, rcp);
    for (int h = 0; h < FHW; ++h)
        for (int i = 0; i < 8; ++i) {
            uint32_t v = fadd((uint32_t)s[i * FHW + h], in[h][i]);
            uint8_t *o = out + 32 * h + 4 * i;
            o[0] = (uint8_t)(v >> 24); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v;
        }
}

/* ======================================================================== */
/* merkle.rs                                                                */
/* ======================================================================== */

/* merkle.rs:30-34: leaf = SHA256(v.to_be_bytes()) */
static void leaf_hash(uint32_t v, uint8_t out[32]) {
    if (g_hash_kind == ORC_HASH_FIELD) {     /* state = (v, 0, ..., 0, 1): the 1 separates leaves from inner nodes */
        uint32_t in[16] = {0};
        in[0] = v % ORC_P; in[15] = 1;
        fh_digest(in, out);
        return;
    }
    uint8_t be[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
    orc_sha256(be, 4, out);
}
/* merkle.rs:42-45: parent = SHA256(left || right) */
static void node_hash(const uint8_t *l, const uint8_t *r, uint8_t out[32]) {
    if (g_hash_kind == ORC_HASH_FIELD) {     /* state = left digest || right digest (8 field elements each) */
        uint32_t in[16];
        for (int i = 0; i < 8; ++i) { in[i] = fh_word(l + 4 * i); in[8 + i] = fh_word(r + 4 * i); }
        fh_digest(in, out);
        return;
    }
    uint8_t cat[64];
    memcpy(cat, l, 32); memcpy(cat + 32, r, 32);
    orc_sha256(cat, 64, out);
}

/* merkle.rs:42-45 for one pair (tests: the levels above a set of device-built nodes) */
void orc_node_hash(const uint8_t *l, const uint8_t *r, uint8_t out[32]) { node_hash(l, r, out); }

/* merkle.rs:14-51 Merkle::new */
int orc_merkle_build(const uint32_t *vals, size_t m, uint8_t *nodes) {
    if (m == 0 || (m & (m - 1))) return -1;          /* merkle.rs:16-21 assert */
    size_t total = 2 * m - 1, offset = total / 2;    /* merkle.rs:27 */
    const int batch = g_hash_kind == ORC_HASH_FIELD && fh_batch_usable();     /* eight field hashes at a time (same digests) */
    if (batch && m >= FHW) {
#pragma omp parallel for schedule(static) if (m >= 4096)
        for (size_t i0 = 0; i0 < m; i0 += FHW) {
            uint32_t in[FHW][FH_T];
            memset(in, 0, sizeof in);
            for (int h = 0; h < FHW; ++h) { in[h][0] = vals[i0 + h] % ORC_P; in[h][15] = 1; }   /* leaf_hash's state */
            fh_digest_batch(in, nodes + 32 * (offset + i0));
        }
    } else {
#pragma omp parallel for schedule(static) if (m >= 4096)
        for (size_t i = 0; i < m; ++i) leaf_hash(vals[i], nodes + 32 * (offset + i));
    }
    while (offset > 0) {                              /* merkle.rs:38-47 */
        offset /= 2;
        if (batch && offset + 1 >= FHW) {
#pragma omp parallel for schedule(static) if (offset >= 2048)
            for (size_t it = 0; it < offset + 1; it += FHW) {
                uint32_t in[FHW][FH_T];
                for (int h = 0; h < FHW; ++h) {
                    const size_t idx = offset + it + h;
                    const uint8_t *lc = nodes + 32 * (2 * idx + 1), *rc = nodes + 32 * (2 * idx + 2);   /* node_hash's state */
                    for (int i = 0; i < 8; ++i) { in[h][i] = fh_word(lc + 4 * i); in[h][8 + i] = fh_word(rc + 4 * i); }
                }
                fh_digest_batch(in, nodes + 32 * (offset + it));
            }
            continue;
        }
#pragma omp parallel for schedule(static) if (offset >= 2048)
        for (size_t it = 0; it < offset + 1; ++it) {
            size_t idx = offset + it;
            node_hash(nodes + 32 * (2 * idx + 1), nodes + 32 * (2 * idx + 2), nodes + 32 * idx);
        }
    }
    return 0;
}
/* merkle.rs:54-71 Merkle::trace */
size_t orc_merkle_trace(const uint8_t *nodes, size_t m, size_t leaf, uint8_t *path_out) {
    size_t i = leaf + (2 * m - 1) / 2, k = 0;
    while (i != 0) {
        if (i % 2 == 0) { memcpy(path_out + 32 * k, nodes + 32 * (i - 1), 32); i -= 2; }
        else            { memcpy(path_out + 32 * k, nodes + 32 * (i + 1), 32); i -= 1; }
        i >>= 1; ++k;
    }
    return k;
}
/* merkle.rs:82-110 compute_root_from_path */
void orc_compute_root_from_path(uint32_t element, size_t index, const uint8_t *path,
                                size_t path_len, uint8_t out[32]) {
    index += ((size_t)1 << path_len) - 1;             /* merkle.rs:84 */
    uint8_t cur[32], nxt[32];
    leaf_hash(element, cur);
    for (size_t k = 0; k < path_len; ++k) {
        if (index % 2 == 0) { node_hash(path + 32 * k, cur, nxt); index -= 2; }
        else                { node_hash(cur, path + 32 * k, nxt); index -= 1; }
        memcpy(cur, nxt, 32);
        index >>= 1;
    }
    memcpy(out, cur, 32);
}

/* ======================================================================== */
/* channel.rs + bincode 1.x default encoding (unpinned)   */
/* ======================================================================== */

void orc_channel_new(orc_channel *c) { memset(c, 0, sizeof *c); }  /* channel.rs:12-17 */
void orc_channel_free(orc_channel *c) { free(c->data); memset(c, 0, sizeof *c); }
/* channel.rs:19-26 commit(): state = SHA256(state || bytes); data += bytes */
void orc_channel_commit_bytes(orc_channel *c, const uint8_t *bytes, size_t n) {
    sha_ctx s; sha_init(&s);
    sha_update(&s, c->state, 32);
    sha_update(&s, bytes, n);
    sha_final(&s, c->state);
    if (c->len + n > c->cap) {
        c->cap = (c->len + n) * 2 + 64;
        c->data = (uint8_t *)realloc(c->data, c->cap);
    }
    memcpy(c->data + c->len, bytes, n);
    c->len += n;
}
static void le32(uint8_t *o, uint32_t v) { o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24); }
static void le64(uint8_t *o, uint64_t v) { for (int i = 0; i < 8; ++i) o[i] = (uint8_t)(v >> (8 * i)); }
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rd64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }
static void commit_u32(orc_channel *c, uint32_t v) { uint8_t b[4]; le32(b, v); orc_channel_commit_bytes(c, b, 4); }
/* channel.rs:28-32 get_u32(): first 4 state bytes big-endian, then committed */
uint32_t orc_channel_get_u32(orc_channel *c) {
    uint32_t f = ((uint32_t)c->state[0] << 24) | ((uint32_t)c->state[1] << 16) |
                 ((uint32_t)c->state[2] << 8) | c->state[3];
    commit_u32(c, f);
    return f;
}
/* (u32, AuthPath): u32 LE, u64 LE count, count*32 raw bytes */
static void commit_val_path(orc_channel *c, uint32_t v, const uint8_t *path, size_t plen) {
    size_t n = 4 + 8 + 32 * plen;
    uint8_t *b = (uint8_t *)malloc(n);
    le32(b, v); le64(b + 4, plen); memcpy(b + 12, path, 32 * plen);
    orc_channel_commit_bytes(c, b, n);
    free(b);
}
/* (u32, u32, AuthPath, AuthPath) */
static void commit_pair_paths(orc_channel *c, uint32_t v0, uint32_t v1, const uint8_t *p0,
                              const uint8_t *p1, size_t plen) {
    size_t n = 8 + 2 * (8 + 32 * plen);
    uint8_t *b = (uint8_t *)malloc(n), *q = b;
    le32(q, v0); q += 4; le32(q, v1); q += 4;
    le64(q, plen); q += 8; memcpy(q, p0, 32 * plen); q += 32 * plen;
    le64(q, plen); q += 8; memcpy(q, p1, 32 * plen);
    orc_channel_commit_bytes(c, b, n);
    free(b);
}

/* ======================================================================== */
/* prover.rs                                                                */
/* ======================================================================== */

/* Multi-query extension (the reference makes one query, prover.rs:263).  With
 * q queries the prover draws q raw indices in a row (q x get_u32) and then commits the openings of
 * each query in turn; q = 1 is byte-identical to the reference's format. */
static uint32_t g_queries = 1;
void orc_set_queries(uint32_t q) { g_queries = q ? (q > 64 ? 64 : q) : 1; }

size_t orc_proof_data_len(uint32_t log_n, uint32_t log_b) {
    size_t L = log_n + log_b, R = log_n;
    size_t per_query = 4 + 4 * (4 + 8 + 32 * L);
    for (size_t i = 0; i < R; ++i) per_query += 8 + 2 * (8 + 32 * (L - i));
    return 32 + 12 + 32 + R * 36 + 4 + g_queries * per_query;
}
/* proof.rs:151-154: size_of::<Proof>() = 32 (state) + 16 (Box<[u8]>) */
size_t orc_proof_size(size_t data_len) { return 48 + data_len; }

/* Shared tail of both modes: prover.rs:254-292 (free term, query, decommit). */
static void decommit(orc_channel *ch, uint32_t log_n, uint32_t log_b, const uint32_t *f_eval,
                     uint8_t *const *trees, uint32_t *const *layers, uint32_t free_term,
                     orc_debug *dbg) {
    size_t n = (size_t)1 << log_n, B = (size_t)1 << log_b, N = n << log_b, R = log_n;
    uint8_t *pa = (uint8_t *)malloc(32 * 64), *pb = (uint8_t *)malloc(32 * 64);
    commit_u32(ch, free_term);                                  /* prover.rs:254 */
    uint32_t qraws[64] = {0};
    for (uint32_t k = 0; k < g_queries; ++k) qraws[k] = orc_channel_get_u32(ch);   /* prover.rs:263 (x q) */
    if (dbg) { dbg->free_term = free_term; dbg->query_raw = qraws[0]; }
    const uint8_t *ftree = trees[0];
    size_t pl;
    for (uint32_t k = 0; k < g_queries; ++k) {
    size_t x = (size_t)qraws[k] % (N - 2 * B);
    pl = orc_merkle_trace(ftree, N, x, pa);         commit_val_path(ch, f_eval[x], pa, pl);         /* :266-274 */
    pl = orc_merkle_trace(ftree, N, x + B, pa);     commit_val_path(ch, f_eval[x + B], pa, pl);     /* :268-275 */
    pl = orc_merkle_trace(ftree, N, x + 2 * B, pa); commit_val_path(ch, f_eval[x + 2 * B], pa, pl); /* :270-276 */
    pl = orc_merkle_trace(trees[1], N, x, pa);      commit_val_path(ch, layers[0][x], pa, pl);      /* :272-277 */
    for (size_t i = 0; i < R; ++i) {                            /* prover.rs:280-289 */
        size_t len = N >> i, xi = x % len, nx = (xi + len / 2) % len;
        size_t p0 = orc_merkle_trace(trees[1 + i], len, xi, pa);
        orc_merkle_trace(trees[1 + i], len, nx, pb);
        commit_pair_paths(ch, layers[i][xi], layers[i][nx], pa, pb, p0);
    }
    }
    free(pa); free(pb);
}

static int prove_ntt(uint32_t log_n, uint32_t log_b, uint32_t a0, uint32_t a1, orc_channel *ch,
                     orc_debug *dbg) {
    size_t n = (size_t)1 << log_n, N = n << log_b, R = log_n;
    uint32_t *trace = (uint32_t *)malloc(n * sizeof(uint32_t));
    orc_trace_fibsq(a0, a1, n - 1, trace);                      /* prover.rs:32-39 */
    uint32_t public_last = trace[n - 2];
    uint32_t *f_eval = (uint32_t *)malloc(N * sizeof(uint32_t));
    orc_lde(trace, log_n, log_b, f_eval);                       /* prover.rs:60-70 */
    uint8_t **trees = (uint8_t **)calloc(R + 2, sizeof(uint8_t *));
    uint32_t **layers = (uint32_t **)calloc(R + 1, sizeof(uint32_t *));
    trees[0] = (uint8_t *)malloc((2 * N - 1) * 32);
    orc_merkle_build(f_eval, N, trees[0]);                      /* prover.rs:81 */
    orc_channel_commit_bytes(ch, trees[0], 32);                 /* prover.rs:85 */
    uint32_t alpha[3];
    for (int i = 0; i < 3; ++i) alpha[i] = orc_channel_get_u32(ch);   /* prover.rs:163-165 */
    layers[0] = (uint32_t *)malloc(N * sizeof(uint32_t));
    orc_compose(f_eval, log_n, log_b, alpha, public_last, layers[0]); /* prover.rs:166-173 */
    trees[1] = (uint8_t *)malloc((2 * N - 1) * 32);
    orc_merkle_build(layers[0], N, trees[1]);                   /* prover.rs:176 */
    orc_channel_commit_bytes(ch, trees[1], 32);                 /* prover.rs:180 */
    if (dbg) {
        memcpy(dbg->alpha_raw, alpha, sizeof alpha);
        dbg->public_last = public_last;
        if (dbg->trace) memcpy(dbg->trace, trace, (n - 1) * sizeof(uint32_t));
        if (dbg->f_eval) memcpy(dbg->f_eval, f_eval, N * sizeof(uint32_t));
        if (dbg->roots) { memcpy(dbg->roots, trees[0], 32); memcpy(dbg->roots + 32, trees[1], 32); }
    }
    for (size_t r = 0; r < R; ++r) {                            /* prover.rs:198-225 */
        size_t m = N >> (r + 1);
        uint32_t beta = orc_channel_get_u32(ch);                /* prover.rs:200 */
        layers[r + 1] = (uint32_t *)malloc(m * sizeof(uint32_t));
        orc_fri_fold_eval(layers[r], log_n, log_b, (uint32_t)r, beta, layers[r + 1]);
        trees[r + 2] = (uint8_t *)malloc((2 * m - 1) * 32);
        orc_merkle_build(layers[r + 1], m, trees[r + 2]);       /* prover.rs:214 */
        orc_channel_commit_bytes(ch, trees[r + 2], 32);         /* prover.rs:224 */
        if (dbg) {
            dbg->beta_raw[r] = beta;
            if (dbg->roots) memcpy(dbg->roots + 32 * (r + 2), trees[r + 2], 32);
        }
    }
    int rc = 0;
    size_t Bsz = (size_t)1 << log_b;                            /* last layer: B equal values */
    for (size_t i = 1; i < Bsz; ++i)
        if (layers[R][i] != layers[R][0]) rc = -10;             /* prover.rs:238 degree 0 */
    if (dbg && dbg->cp_layers) {
        size_t off = 0;
        for (size_t r = 0; r <= R; ++r) { memcpy(dbg->cp_layers + off, layers[r], (N >> r) * sizeof(uint32_t)); off += N >> r; }
    }
    decommit(ch, log_n, log_b, f_eval, trees, layers, layers[R][0], dbg);
    for (size_t r = 0; r < R + 2; ++r) free(trees[r]);
    for (size_t r = 0; r <= R; ++r) free(layers[r]);
    free(trees); free(layers); free(f_eval); free(trace);
    return rc;
}

/* Literal prover.rs with polynomial.rs arithmetic.  O(n^3): n <= 2^10 only. */
static int prove_naive(uint32_t log_n, uint32_t log_b, uint32_t a0v, uint32_t a1v, orc_channel *ch,
                       orc_debug *dbg) {
    const uint32_t p = ORC_P;
    size_t n = (size_t)1 << log_n, N = n << log_b, R = log_n;
    int rc = 0;
    uint32_t *a = (uint32_t *)malloc(n * sizeof(uint32_t));
    orc_trace_fibsq(a0v, a1v, n - 1, a);                        /* prover.rs:32-39 */
    uint32_t w = orc_generator(p);                              /* prover.rs:45 */
    uint32_t gg = orc_pow(w, (p - 1) >> log_n, p);              /* prover.rs:48 */
    uint32_t hh = orc_pow(w, (p - 1) >> (log_n + log_b), p);    /* prover.rs:49 */
    if (log_n <= 12 && (orc_order(gg, p) != n || orc_order(hh, p) != N)) rc = -1; /* :52-53 */
    uint32_t *g = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *dom = (uint32_t *)malloc(N * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) g[i] = orc_pow(gg, (uint32_t)i, p);           /* :56 */
    for (size_t i = 0; i < N; ++i) dom[i] = orc_mul(w, orc_pow(hh, (uint32_t)i, p), p); /* :57,:69 */
    poly f_poly = lagrange_naive(g, a, n - 1, p);               /* prover.rs:60-61 */
    for (size_t i = 0; i + 1 < n; ++i)                          /* prover.rs:64-66 */
        if (poly_solve(&f_poly, g[i], p) != a[i]) rc = -2;
    uint32_t *f_eval = (uint32_t *)malloc(N * sizeof(uint32_t));
    for (size_t i = 0; i < N; ++i) f_eval[i] = poly_solve(&f_poly, dom[i], p);   /* :70 */
    uint8_t **trees = (uint8_t **)calloc(R + 2, sizeof(uint8_t *));
    uint32_t **layers = (uint32_t **)calloc(R + 1, sizeof(uint32_t *));
    trees[0] = (uint8_t *)malloc((2 * N - 1) * 32);
    orc_merkle_build(f_eval, N, trees[0]);                      /* :81 */
    orc_channel_commit_bytes(ch, trees[0], 32);                 /* :85 */

    /* constraints, prover.rs:101-145 */
    poly c0, c0r, c1, c1r, c2, c2r, den, t2r;
    {
        poly k = poly_monomial(a[0], 0), num = poly_sub(&f_poly, &k, p), d = poly_linear(g[0], p);
        if (poly_div(&num, &d, &c0, &c0r, p)) rc = -3;
        poly_free(&k); poly_free(&num); poly_free(&d);
    }
    {
        poly k = poly_monomial(a[n - 2], 0), num = poly_sub(&f_poly, &k, p), d = poly_linear(g[n - 2], p);
        if (poly_div(&num, &d, &c1, &c1r, p)) rc = -3;
        poly_free(&k); poly_free(&num); poly_free(&d);
    }
    {
        poly t0 = poly_clone(&f_poly); poly_apply_const(&t0, g[2], p);          /* :134 */
        poly t1 = poly_clone(&f_poly); poly_apply_const(&t1, g[1], p);          /* :135 */
        poly t1s = poly_mul(&t1, &t1, p);                                        /* :136 */
        poly t2 = poly_mul(&f_poly, &f_poly, p);                                 /* :137 */
        poly s = poly_sub(&t0, &t1s, p), num = poly_sub(&s, &t2, p);            /* :138 */
        poly xn = poly_monomial(1, n), one = poly_monomial(1, 0);
        poly d0 = poly_sub(&xn, &one, p);                                        /* :140 */
        poly tp0 = poly_linear(g[n - 3], p), tp1 = poly_linear(g[n - 2], p), tp2 = poly_linear(g[n - 1], p);
        poly m20 = poly_mul(&tp2, &tp0, p), m = poly_mul(&m20, &tp1, p);         /* :144 */
        if (poly_div(&d0, &m, &den, &t2r, p)) rc = -3;
        if (poly_div(&num, &den, &c2, &c2r, p)) rc = -3;                         /* :145 */
        poly_free(&t0); poly_free(&t1); poly_free(&t1s); poly_free(&t2); poly_free(&s); poly_free(&num);
        poly_free(&xn); poly_free(&one); poly_free(&d0); poly_free(&tp0); poly_free(&tp1); poly_free(&tp2);
        poly_free(&m20); poly_free(&m);
    }
    if (c0r.len || c1r.len || t2r.len || c2r.len) rc = -4;      /* prover.rs:148-151 */
    if (c0.len != n - 2 || c1.len != n - 2 || c2.len != n) rc = -5;  /* prover.rs:154-156 */
    uint32_t alpha[3];
    for (int i = 0; i < 3; ++i) alpha[i] = orc_channel_get_u32(ch);  /* prover.rs:163-165 */
    poly cp_poly;
    {
        poly s0 = poly_clone(&c0), s1 = poly_clone(&c1), s2 = poly_clone(&c2);
        poly_scale(&s0, orc_from_u32(alpha[0], p), p);
        poly_scale(&s1, orc_from_u32(alpha[1], p), p);
        poly_scale(&s2, orc_from_u32(alpha[2], p), p);
        poly s01 = poly_add(&s0, &s1, p);
        cp_poly = poly_add(&s01, &s2, p);                       /* prover.rs:166 */
        poly_free(&s0); poly_free(&s1); poly_free(&s2); poly_free(&s01);
    }
    if (cp_poly.len != n) rc = -6;                              /* prover.rs:169 */
    layers[0] = (uint32_t *)malloc(N * sizeof(uint32_t));
    for (size_t i = 0; i < N; ++i) layers[0][i] = poly_solve(&cp_poly, dom[i], p);  /* :173 */
    trees[1] = (uint8_t *)malloc((2 * N - 1) * 32);
    orc_merkle_build(layers[0], N, trees[1]);                   /* :176 */
    orc_channel_commit_bytes(ch, trees[1], 32);                 /* :180 */
    if (dbg) {
        memcpy(dbg->alpha_raw, alpha, sizeof alpha);
        dbg->public_last = a[n - 2];
        dbg->cp_degree = (uint32_t)(cp_poly.len - 1);
        if (dbg->trace) memcpy(dbg->trace, a, (n - 1) * sizeof(uint32_t));
        if (dbg->f_eval) memcpy(dbg->f_eval, f_eval, N * sizeof(uint32_t));
        if (dbg->roots) { memcpy(dbg->roots, trees[0], 32); memcpy(dbg->roots + 32, trees[1], 32); }
    }
    /* FRI, prover.rs:198-225 */
    poly cur = cp_poly;
    size_t dlen = N;
    for (size_t r = 0; r < R; ++r) {
        uint32_t beta = orc_channel_get_u32(ch);                /* :200 */
        poly nxt = poly_new(cur.len / 2);
        orc_fri_coef_fold(cur.c, cur.len, orc_from_u32(beta, p), nxt.c, p);      /* :201 */
        dlen /= 2;                                              /* :204-208 */
        for (size_t i = 0; i < dlen; ++i) dom[i] = orc_pow(dom[i], 2, p);
        layers[r + 1] = (uint32_t *)malloc(dlen * sizeof(uint32_t));
        for (size_t i = 0; i < dlen; ++i) layers[r + 1][i] = poly_solve(&nxt, dom[i], p);  /* :211 */
        trees[r + 2] = (uint8_t *)malloc((2 * dlen - 1) * 32);
        orc_merkle_build(layers[r + 1], dlen, trees[r + 2]);    /* :214 */
        orc_channel_commit_bytes(ch, trees[r + 2], 32);         /* :224 */
        if (nxt.len != (n >> (r + 1))) rc = -7;                 /* :228-238 */
        if (dbg) {
            dbg->beta_raw[r] = beta;
            if (dbg->roots) memcpy(dbg->roots + 32 * (r + 2), trees[r + 2], 32);
        }
        poly_free(&cur);
        cur = nxt;
    }
    if (dbg && dbg->cp_layers) {
        size_t off = 0;
        for (size_t r = 0; r <= R; ++r) { memcpy(dbg->cp_layers + off, layers[r], (N >> r) * sizeof(uint32_t)); off += N >> r; }
    }
    uint32_t free_term = cur.len ? cur.c[cur.len - 1] : 0;      /* prover.rs:254: poly[0] = leading */
    decommit(ch, log_n, log_b, f_eval, trees, layers, free_term, dbg);
    poly_free(&cur); poly_free(&f_poly);
    poly_free(&c0); poly_free(&c0r); poly_free(&c1); poly_free(&c1r); poly_free(&c2); poly_free(&c2r);
    poly_free(&den); poly_free(&t2r);
    for (size_t r = 0; r < R + 2; ++r) free(trees[r]);
    for (size_t r = 0; r <= R; ++r) free(layers[r]);
    free(trees); free(layers); free(f_eval); free(a); free(g); free(dom);
    return rc;
}

int orc_prove(uint32_t log_n, uint32_t log_b, uint32_t a0, uint32_t a1, int mode,
              uint8_t *proof_out, size_t cap, size_t *proof_len, uint8_t final_state[32],
              orc_debug *dbg) {
    if (log_n < 2 || log_b < 1 || log_n + log_b > 30) return -100;
    if (mode == ORC_MODE_NAIVE && log_n > 10) return -101;
    /* n = 8 is degenerate: g^4 = -1 makes the leading terms of f(gx)^2 + f(x)^2 cancel, so c2 has
     * degree < n-1 and the generalised asserts of prover.rs:156 / :169 fail (the reference would panic). */
    if (log_n == 3) return -103;
    orc_channel ch; orc_channel_new(&ch);                       /* main.rs:19 */
    int rc = mode == ORC_MODE_NAIVE ? prove_naive(log_n, log_b, a0, a1, &ch, dbg)
                                    : prove_ntt(log_n, log_b, a0, a1, &ch, dbg);
    if (proof_len) *proof_len = ch.len;
    if (proof_out) { if (ch.len <= cap) memcpy(proof_out, ch.data, ch.len); else rc = -102; }
    if (final_state) memcpy(final_state, ch.state, 32);         /* channel.rs:34-36 */
    orc_channel_free(&ch);
    return rc;
}

/* generate_proof(channel) (prover.rs:9) on a channel that already holds a committed prefix: the caller's
 * channel is an argument of the reference's prover, so whatever it absorbed before binds the proof.
 * `prefix` is committed first (one Channel::commit, channel.rs:19-26), then the prover runs on the same
 * channel; proof_out receives the whole Channel.data (prefix included, channel.rs:34-36). */
int orc_prove_prefixed(const uint8_t *prefix, size_t prefix_len, uint32_t log_n, uint32_t log_b, uint32_t a0, uint32_t a1,
                       uint8_t *proof_out, size_t cap, size_t *proof_len, uint8_t final_state[32]) {
    if (log_n < 2 || log_b < 1 || log_n + log_b > 30 || log_n == 3) return -100;
    orc_channel ch; orc_channel_new(&ch);
    if (prefix_len) orc_channel_commit_bytes(&ch, prefix, prefix_len);
    int rc = prove_ntt(log_n, log_b, a0, a1, &ch, NULL);
    if (proof_len) *proof_len = ch.len;
    if (proof_out) { if (ch.len <= cap) memcpy(proof_out, ch.data, ch.len); else rc = -102; }
    if (final_state) memcpy(final_state, ch.state, 32);
    orc_channel_free(&ch);
    return rc;
}

/* ======================================================================== */
/* proof.rs: verify()                                                       */
/* ======================================================================== */

typedef struct { const uint8_t *p; size_t left; int bad; } rdr;
static const uint8_t *take(rdr *r, size_t n) {
    if (r->left < n) { r->bad = 1; return NULL; }
    const uint8_t *q = r->p; r->p += n; r->left -= n; return q;
}
static uint32_t take_u32(rdr *r) { const uint8_t *q = take(r, 4); return q ? rd32(q) : 0; }
static const uint8_t *take_path(rdr *r, size_t *plen) {
    const uint8_t *q = take(r, 8);
    if (!q) return NULL;
    uint64_t c = rd64(q);
    if (c > 64) { r->bad = 1; return NULL; }
    *plen = (size_t)c;
    return take(r, 32 * (size_t)c);
}

int orc_verify(const uint8_t *data, size_t len, uint32_t log_n, uint32_t log_b, uint32_t public_last) {
    size_t n = (size_t)1 << log_n, B = (size_t)1 << log_b, N = n << log_b, R = log_n;
    rdr r = {data, len, 0};
    /* proof.rs:20-46 */
    const uint8_t *f_root = take(&r, 32);
    uint32_t alpha0 = take_u32(&r), alpha1 = take_u32(&r), alpha2 = take_u32(&r);
    const uint8_t *roots[40];
    uint32_t betas[40];
    roots[0] = take(&r, 32); betas[0] = 0;
    for (size_t i = 0; i < R; ++i) { betas[i + 1] = take_u32(&r); roots[i + 1] = take(&r, 32); }
    uint32_t free_term = take_u32(&r);
    uint32_t test_raws[64];
    for (uint32_t k = 0; k < g_queries; ++k) test_raws[k] = take_u32(&r);
    for (uint32_t qk = 0; qk < g_queries; ++qk) {
    uint32_t test_raw = test_raws[qk];
    uint32_t fv[4]; const uint8_t *fp[4]; size_t fpl[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) { fv[i] = take_u32(&r); fp[i] = take_path(&r, &fpl[i]); }
    uint32_t lx[40], lnx[40]; const uint8_t *lpx[40], *lpnx[40]; size_t lplx[40], lplnx[40];
    for (size_t i = 0; i < R; ++i) {
        lx[i] = take_u32(&r); lnx[i] = take_u32(&r);
        lplx[i] = lplnx[i] = 0;
        lpx[i] = take_path(&r, &lplx[i]); lpnx[i] = take_path(&r, &lplnx[i]);
    }
    if (r.bad) return -1;                                       /* bincode unwrap() panics */
    /* proof.rs:49-54 */
    uint32_t w = gen_w(), g = gen_of_order_log(log_n), h = gen_of_order_log(log_n + log_b);
    size_t tp = (size_t)test_raw % (N - 2 * B);                 /* proof.rs:60 */
    uint32_t x = fmul(w, fpow(h, tp));
    {   /* proof.rs:63-77 */
        uint32_t f_x = fv[0] % ORC_P, f_gx = fv[1] % ORC_P, f_ggx = fv[2] % ORC_P;
        uint32_t gm1 = finv(g), gm2 = fmul(gm1, gm1), gm3 = fmul(gm2, gm1);
        uint32_t p0 = fmul(fsub(f_x, 1), finv(fsub(x, 1)));
        uint32_t p1 = fmul(fsub(f_x, public_last % ORC_P), finv(fsub(x, gm2)));
        uint32_t num = fsub(fsub(f_ggx, fmul(f_gx, f_gx)), fmul(f_x, f_x));
        uint32_t den = fmul(fsub(fpow(x, n), 1),
                            finv(fmul(fmul(fsub(x, gm3), fsub(x, gm2)), fsub(x, gm1))));
        uint32_t p2 = fmul(num, finv(den));
        uint32_t cp0 = fadd(fadd(fmul(alpha0 % ORC_P, p0), fmul(alpha1 % ORC_P, p1)), fmul(alpha2 % ORC_P, p2));
        if (cp0 != fv[3]) return -2;                            /* proof.rs:76 */
    }
    uint8_t root[32];
    size_t L = log_n + log_b;
    /* proof.rs:80-95 */
    if (fpl[0] != L || fpl[1] != L || fpl[2] != L || fpl[3] != L) return -3;
    orc_compute_root_from_path(fv[0], tp, fp[0], fpl[0], root);         if (memcmp(root, f_root, 32)) return -4;
    orc_compute_root_from_path(fv[1], tp + B, fp[1], fpl[1], root);     if (memcmp(root, f_root, 32)) return -5;
    orc_compute_root_from_path(fv[2], tp + 2 * B, fp[2], fpl[2], root); if (memcmp(root, f_root, 32)) return -6;
    orc_compute_root_from_path(fv[3], tp, fp[3], fpl[3], root);         if (memcmp(root, roots[0], 32)) return -7;
    /* proof.rs:101-126: fold checks for layers 0..R-2, then layer R-1 against the free term */
    uint32_t inv2 = finv(2);
    for (size_t k = 0; k < R; ++k) {
        uint32_t xk = fpow(x, (uint64_t)1 << k);
        uint32_t gx = fmul(fadd(lx[k] % ORC_P, lnx[k] % ORC_P), inv2);
        uint32_t hx = fmul(fsub(lx[k] % ORC_P, lnx[k] % ORC_P), finv(fmul(xk, 2)));
        uint32_t calc = fadd(gx, fmul(betas[k + 1] % ORC_P, hx));
        uint32_t expect = (k + 1 < R) ? lx[k + 1] : free_term;
        if (calc != expect) return -(int)(100 + k);
    }
    /* proof.rs:129-148 */
    for (size_t k = 0; k < R; ++k) {
        size_t size = N >> k;
        if (lplx[k] != L - k || lplnx[k] != L - k) return -(int)(200 + k);
        orc_compute_root_from_path(lx[k], tp % size, lpx[k], lplx[k], root);
        if (memcmp(root, roots[k], 32)) return -(int)(300 + k);
        orc_compute_root_from_path(lnx[k], (tp + size / 2) % size, lpnx[k], lplnx[k], root);
        if (memcmp(root, roots[k], 32)) return -(int)(400 + k);
    }
    }
    if (r.left != 0) return -8;
    return 0;
}
