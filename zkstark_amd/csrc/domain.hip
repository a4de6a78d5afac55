// domain.hip -- coset evaluation domains and the transforms over them (host side): power tables, the
// transform plan, LDE / composition / fold launch sequences, and the error plumbing of the C ABI.
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <new>

#include "internal.hpp"

namespace zk {
namespace impl {

namespace {
thread_local std::string g_last_error;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
const char* last_error() { return g_last_error.c_str(); }

Plan make_plan(uint32_t log_m) {
    Plan p;
    uint32_t np = (log_m + kMaxRadixLog - 1) / kMaxRadixLog;
    if (np == 0) np = 1;
    uint32_t base = log_m / np, extra = log_m % np;
    p.nd = np;
    for (uint32_t d = 0; d < np; ++d) p.bits[d] = base + (d < extra ? 1 : 0);
    return p;
}

uint32_t pick_logC(uint32_t log_total, uint32_t logR) {
    uint32_t cols = log_total - logR, cap = ntt_pass_tile_log(log_total, logR) - logR;
    return cols < cap ? cols : cap;
}

// ---- power tables --------------------------------------------------------------
int build_table(uint32_t root, uint32_t log_order, DevTable* t) {
    uint32_t lo_bits = (log_order + 1) / 2, hi_bits = log_order - lo_bits;
    std::vector<uint32_t> lo((size_t)1 << lo_bits), hi((size_t)1 << hi_bits);
    uint32_t acc = 1;
    for (size_t j = 0; j < lo.size(); ++j) { lo[j] = to_mont(acc); acc = mulmod(acc, root); }
    uint32_t step = powmod(root, (uint64_t)1 << lo_bits);
    acc = 1;
    for (size_t j = 0; j < hi.size(); ++j) { hi[j] = to_mont(acc); acc = mulmod(acc, step); }
    HIPCHK(hipMalloc(&t->lo, lo.size() * 4));
    HIPCHK(hipMalloc(&t->hi, hi.size() * 4));
    HIPCHK(hipMemcpy(t->lo, lo.data(), lo.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t->hi, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    t->lo_bits = lo_bits;
    return ZK_OK;
}
void free_table(DevTable* t) {
    if (t->lo) (void)hipFree(t->lo);
    if (t->hi) (void)hipFree(t->hi);
    t->lo = t->hi = nullptr;
}

// Inverse transform, natural order in, digit-reversed out, optionally scaled by scale_mont.
int run_dif(const uint32_t* src, uint32_t* data, uint32_t log_m, const Plan& pl, PowTable tw_inv, uint32_t L, uint32_t scale_mont, hipStream_t s, Profiler* prof,
            uint32_t batch, size_t src_stride, size_t data_stride) {
    uint32_t inner = log_m;
    for (uint32_t d = 0; d < pl.nd; ++d) {
        inner -= pl.bits[d];
        NttPassArgs a{};
        a.batch = batch; a.src_stride = d == 0 ? src_stride : data_stride; a.dst_stride = data_stride;
        a.src = d == 0 ? src : data; a.dst = data; a.log_total = log_m;
        a.logR = pl.bits[d]; a.logS = inner; a.logC = pick_logC(log_m, a.logR); a.tile_log = ntt_pass_tile_log(log_m, a.logR);
        a.L = L; a.tw = tw_inv; a.scale_mont = (inner == 0) ? scale_mont : 0;
        HIPCHK(launch_ntt_pass(a, NTT_DIF, s, prof));
    }
    return ZK_OK;
}
// Forward transform, digit-reversed in, natural out.
int run_dit(uint32_t* data, uint32_t log_m, const Plan& pl, PowTable tw, uint32_t L, hipStream_t s) {
    uint32_t inner = 0;
    for (int d = (int)pl.nd - 1; d >= 0; --d) {
        NttPassArgs a{};
        a.src = data; a.dst = data; a.log_total = log_m;
        a.logR = pl.bits[d]; a.logS = inner; a.logC = pick_logC(log_m, a.logR); a.tile_log = ntt_pass_tile_log(log_m, a.logR);
        a.L = L; a.tw = tw;
        HIPCHK(launch_ntt_pass(a, NTT_DIT, s));
        inner += pl.bits[d];
    }
    return ZK_OK;
}



void dom_free(zk_dom* d) {
    if (!d) return;
    free_table(&d->H); free_table(&d->Hinv); free_table(&d->W);
    if (d->d_inv_xm1) (void)hipFree(d->d_inv_xm1);
    delete d;
}

// fold_only: tables for fri_fold only (no w-power table, no 1/(x-1) table)
int dom_make(int device, uint32_t log_n, uint32_t log_b, uint32_t shift, bool fold_only, hipStream_t stream, zk_dom** out) {
    *out = nullptr;
    if (log_n < 1 || log_b > 5 || log_n + log_b > 30 || log_n + log_b < 1)
        return fail(ZK_ERR_INVALID, "domain: need 1 <= log_n, log_blowup <= 5, log_n + log_blowup <= 30 (got %u, %u)", log_n, log_b);
    if (shift == 0 || shift >= P) return fail(ZK_ERR_INVALID, "domain: shift must be a non-zero canonical residue");
    HIPCHK(hipSetDevice(device));
    zk_dom* d = new (std::nothrow) zk_dom();
    if (!d) return fail(ZK_ERR_NOMEM, "out of host memory");
    d->device = device;
    d->log_n = log_n; d->log_b = log_b; d->L = log_n + log_b;
    d->n = (size_t)1 << log_n; d->B = (size_t)1 << log_b; d->N = d->n << log_b;
    d->shift = shift;
    d->g = root_of_unity(log_n);
    d->h = root_of_unity(d->L);
    d->plan = make_plan(log_n);
    int rc;
    if ((rc = build_table(d->h, d->L, &d->H)) || (rc = build_table(invmod(d->h), d->L, &d->Hinv))) { dom_free(d); return rc; }
    uint32_t gm1 = invmod(d->g);
    d->shift_mont = to_mont(shift);
    d->gm1_mont = to_mont(gm1);
    d->gm2_mont = to_mont(mulmod(gm1, gm1));
    d->gm3_mont = to_mont(mulmod(mulmod(gm1, gm1), gm1));
    d->ninv_mont = to_mont(invmod((uint32_t)(d->n % P)));
    d->inv2_mont = to_mont(invmod(2));
    {
        const uint32_t inv2 = invmod(2);
        uint32_t wr = shift;                                     // w^(2^r)
        for (uint32_t r = 0; r < d->L && r < 32; ++r) { d->fold_k[r] = mulmod(invmod(wr), inv2); wr = mulmod(wr, wr); }
        uint32_t xn = powmod(shift, d->n);                       // x^n on the domain takes B values: shift^n (h^n)^(i mod B)
        const uint32_t hn = powmod(d->h, d->n);
        for (size_t r = 0; r < d->B; ++r) { const uint32_t den = sub(xn, 1); d->inv_den[r] = den ? invmod(den) : 0u; xn = mulmod(xn, hn); }
    }
    if (!fold_only) {
        if ((rc = build_table(shift, log_n, &d->W))) { dom_free(d); return rc; }
        // x_i - 1 must be invertible on the whole domain: shift^N != 1
        if (powmod(shift, d->N) == 1) { dom_free(d); return fail(ZK_ERR_INVALID, "domain: shift lies in the evaluation subgroup"); }
        hipError_t e = hipMalloc((void**)&d->d_inv_xm1, d->N * 4);
        if (e != hipSuccess) { dom_free(d); return fail(ZK_ERR_NOMEM, "hipMalloc(%zu) failed", d->N * 4); }
        d->device_bytes += d->N * 4;
        e = launch_build_inv_xm1(d->d_inv_xm1, d->L, d->H.view(), d->shift_mont, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { dom_free(d); return fail(ZK_ERR_HIP, "inv_xm1 build failed: %s", hipGetErrorString(e)); }
    }
    *out = d;
    return ZK_OK;
}

// lagrange + solve over the coset (polynomial.rs:337, :49; prover.rs:60-70).
// d_trace: n words (a[0..n-2], 0); d_coef: 2n words scratch; d_out: N words.
// batch > 1: `batch` independent traces, proof b at d_trace + b*n, d_coef + b*2n, d_out + b*N.
int dom_lde(const zk_dom* d, const uint32_t* d_trace, uint32_t* d_coef, uint32_t* d_out, hipStream_t s, Profiler* prof, uint32_t batch) {
    const size_t ts = batch > 1 ? d->n : 0, cs = batch > 1 ? 2 * d->n : 0, os = batch > 1 ? d->N : 0;
    // iNTT_g of (a_0 .. a_{n-2}, 0): natural -> digit-reversed, unscaled (1/n is folded into the next pass)
    int rc = run_dif(d_trace, d_coef, d->log_n, d->plan, d->Hinv.view(), d->L, 0, s, prof, batch, ts, cs);
    if (rc) return rc;
    // virtual-point correction, coset shift and 1/n (coef_prepare): applied by the first LDE pass, which loads the
    // coefficient blocks of its tile once, prepares them and keeps them in LDS for the B columns that use them; only
    // shapes the register-radix kernel does not take run the separate sweep d_coef[0..n) -> d_coef[n..2n)
    uint32_t* d_prep = d_coef + d->n;
    CoefPrepArgs pa{};
    pa.log_n = d->log_n; pa.log_b = d->log_b;
    pa.tw = d->H.view(); pa.wtab = d->W.view(); pa.ninv_mont = d->ninv_mont;
    pa.nd = d->plan.nd;
    for (uint32_t t = 0; t < d->plan.nd; ++t) pa.dig_bits[t] = d->plan.bits[t];
    // size-N forward transform of the zero-padded coefficients
    uint32_t inner = d->log_b;
    for (int q = (int)d->plan.nd - 1; q >= 0; --q) {
        NttPassArgs a{};
        a.log_total = d->L; a.logR = d->plan.bits[q]; a.logS = inner; a.logC = pick_logC(d->L, a.logR); a.tile_log = ntt_pass_tile_log(d->L, a.logR);
        a.L = d->L; a.tw = d->H.view();
        a.dst = d_out;
        a.batch = batch; a.dst_stride = os; a.src_stride = os;
        if (q == (int)d->plan.nd - 1) {
            if (a.logC < d->log_b) a.logC = d->log_b;
            a.src_stride = cs;
            const bool fuse = ntt_fast_ok(a, NTT_DIT_LDE);
            if (fuse) {
                a.src = d_coef;                                   // raw DIF output: prepared at the load
                a.prep = 1; a.prep_log_n = d->log_n; a.prep_log_b = d->log_b; a.prep_ninv_mont = d->ninv_mont;
                a.prep_nd = d->plan.nd; a.prep_wtab = d->W.view();
                for (uint32_t t = 0; t < d->plan.nd; ++t) a.prep_bits[t] = d->plan.bits[t];
            } else {
                HIPCHK(launch_coef_prepare(d_coef, d_prep, pa, s, prof, batch, cs, cs));
                a.src = d_prep;
            }
            HIPCHK(launch_ntt_pass(a, NTT_DIT_LDE, s, prof));
        } else {
            a.src = d_out;
            HIPCHK(launch_ntt_pass(a, NTT_DIT, s, prof));
        }
        inner += d->plan.bits[q];
    }
    return ZK_OK;
}

// prover.rs:101-173 pointwise on the domain.
int compose_args(const zk_dom* d, const uint32_t* d_f, uint32_t* d_cp, uint32_t first, uint32_t last,
                 const uint32_t alpha_raw[3], ComposeArgs& a) {
    a = ComposeArgs{};
    a.f = d_f; a.inv_xm1 = d->d_inv_xm1; a.cp = d_cp;
    a.logN = d->L; a.log_b = d->log_b;
    a.htab = d->H.view();
    a.w_mont = d->shift_mont; a.gm1_mont = d->gm1_mont; a.gm2_mont = d->gm2_mont; a.gm3_mont = d->gm3_mont;
    a.first = first % P; a.last = last % P;
    uint32_t a0 = alpha_raw[0] % P, a1 = alpha_raw[1] % P, a2 = alpha_raw[2] % P;   // field.rs:20-24
    a.alpha0_mont = to_mont(a0);
    a.alpha1g2_mont = to_mont(mulmod(a1, mulmod(d->g, d->g)));
    for (size_t r = 0; r < d->B; ++r) {
        if (d->inv_den[r] == 0) return fail(ZK_ERR_INVALID, "compose: x^n = 1 on the domain");
        a.zz[r] = to_mont(to_mont(mulmod(a2, d->inv_den[r])));
    }
    return ZK_OK;
}
int dom_compose(const zk_dom* d, const uint32_t* d_f, uint32_t* d_cp, uint32_t first, uint32_t last,
                const uint32_t alpha_raw[3], hipStream_t s, Profiler* prof) {
    ComposeArgs a;
    int rc = compose_args(d, d_f, d_cp, first, last, alpha_raw, a);
    if (rc) return rc;
    HIPCHK(launch_compose(a, s, prof));
    return ZK_OK;
}

// polynomial.rs:385-400 + prover.rs:204-211 in evaluation form: layer of 2^log_m values at
// x_i = (shift h^i)^(2^round) -> 2^(log_m-1) values.
int fold_args(const zk_dom* d, const uint32_t* d_in, uint32_t* d_out, uint32_t log_m, uint32_t round, uint32_t beta_raw, FoldArgs& a) {
    if (log_m < 1 || log_m + round != d->L) return fail(ZK_ERR_INVALID, "fold: layer size 2^%u does not match round %u of a 2^%u domain", log_m, round, d->L);
    a = FoldArgs{};
    a.in = d_in; a.out = d_out; a.log_m = log_m; a.round = round;
    a.hinv = d->Hinv.view(); a.L = d->L;
    a.inv2_mont = d->inv2_mont;
    a.c_mont = to_mont(mulmod(beta_raw % P, d->fold_k[round]));   // beta * w^(-2^r) / 2
    return ZK_OK;
}
int dom_fold(const zk_dom* d, const uint32_t* d_in, uint32_t* d_out, uint32_t log_m, uint32_t round, uint32_t beta_raw,
             hipStream_t s, Profiler* prof) {
    FoldArgs a;
    int rc = fold_args(d, d_in, d_out, log_m, round, beta_raw, a);
    if (rc) return rc;
    HIPCHK(launch_fri_fold(a, s, prof));
    return ZK_OK;
}


// Waits until the mailbox carries the sequence number of the last commit launch (polling host-coherent
// memory: no blit kernel, no stream synchronisation on the commit -> challenge path).
int wait_flag(const uint32_t* flag, uint32_t want, hipStream_t stream, int (*poll)(void*), void* poll_user, double timeout_s) {
    auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) {
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
        __builtin_ia32_pause();                          // be kind to the sibling hardware thread
#endif
        if (poll && (++spins & 0xFFF) == 0) {
            if (int prc = poll(poll_user)) return prc;
        } else if (!poll) {
            ++spins;
        }
        if ((spins & 0xFFFF) == 0) {
            hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != want)
                return fail(ZK_ERR_HIP, "merkle digests were never posted (stream drained)");
            if (q != hipSuccess && q != hipErrorNotReady)
                return fail(ZK_ERR_HIP, "device error while waiting for merkle digests: %s", hipGetErrorString(q));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                return fail(ZK_ERR_HIP, "timed out after %.0f s waiting for merkle digests", timeout_s);
        }
    }
    return ZK_OK;
}

double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// merkle.rs:54-71: node indices of the authentication path of `leaf` in a tree of m leaves
void path_nodes(size_t m, size_t leaf, std::vector<size_t>& out) {
    size_t i = leaf + (2 * m - 1) / 2;
    while (i != 0) {
        if (i % 2 == 0) { out.push_back(i - 1); i -= 2; }
        else { out.push_back(i + 1); i -= 1; }
        i >>= 1;
    }
}

}  // namespace impl
}  // namespace zk
