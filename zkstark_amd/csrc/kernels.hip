// kernels.hip -- hand-written gfx950 kernels for the STARK-101 prover hot path.
//
//   ntt_pass_kernel      LDS-tiled radix-2^k NTT passes (replaces polynomial.rs:337-383 lagrange
//                        and polynomial.rs:49-56 solve as used at prover.rs:60-70)
//   compose_kernel       pointwise constraint composition (prover.rs:101-173)
//   fri_fold_kernel      evaluation-form FRI fold (polynomial.rs:385-400 + prover.rs:204-211)
//   merkle_*_kernel      SHA-256 Merkle heap (merkle.rs:14-51)
//   gather_kernel        decommitment gather (merkle.rs:54-71, prover.rs:266-289)
//
// All bulk data is canonical u32 residues; all constants are in Montgomery form (field.hpp).
// No MFMA anywhere: butterflies and SHA rounds are not a dense contraction.
#include "kernels.hpp"

#include <cstdlib>

#include "field.hpp"
#include "fieldhash_f64.hpp"
#include "sha256_quad.hpp"
#include "sha256.hpp"

namespace zk {

__device__ __forceinline__ uint32_t pow_lookup(const PowTable& t, uint32_t e) {
    return mont_mul(t.hi[e >> t.lo_bits], t.lo[e & ((1u << t.lo_bits) - 1u)]);
}

constexpr uint32_t R2_MONT = R2;   // 2^64 mod P (fieldhash.hpp): canonical -> Montgomery

// Montgomery-domain inverse by Fermat: a^(P-2), P-2 = 0xBFFFFFFF.
__device__ uint32_t mont_inv(uint32_t a) {
    uint32_t r = R1, b = a;
    uint32_t e = P - 2;
#pragma unroll 1
    for (int i = 0; i < 32; ++i) {
        if (e & 1) r = mont_mul(r, b);
        b = mont_mul(b, b);
        e >>= 1;
    }
    return r;
}

// ===========================================================================
// NTT passes
// ===========================================================================
//
// Generic pass for any radix and size (the full-size tiles with R = 64/128/256 take the
// register-radix kernel of ntt_fast.hip instead; this one serves small transforms and odd radices).
// The array is viewed as [A][R][S] with S fastest.  A workgroup owns a tile of C
// columns (a, s) and all R rows t of those columns, stages it in LDS with row
// pitch C+1, runs the log2(R) radix-2 stages there and writes the tile back in
// place.  Global accesses are rows of C contiguous words (C*4 >= 128 B whenever
// S >= 32), or one fully contiguous block when S < C.
//
// Inverse transforms run decimation-in-frequency (natural -> digit-reversed
// order), forward transforms decimation-in-time (digit-reversed -> natural), with
// the same radix list, so iNTT followed by NTT never needs a permutation pass.
// Bit reversal inside one radix-R block is absorbed into the LDS row index.

constexpr int kNttThreads = 256;

__device__ __forceinline__ void tile_coords(uint32_t l, uint32_t logR, uint32_t logS, uint32_t logC,
                                            uint32_t& t, uint32_t& c) {
    if (logS >= logC) {            // row of C contiguous words
        t = l >> logC;
        c = l & ((1u << logC) - 1u);
    } else {                       // whole tile is one contiguous block
        uint32_t s = l & ((1u << logS) - 1u);
        t = (l >> logS) & ((1u << logR) - 1u);
        c = ((l >> (logR + logS)) << logS) | s;
    }
}

template <uint32_t MODE>
__global__ __launch_bounds__(kNttThreads) void ntt_pass_kernel(NttPassArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    p.src += (size_t)blockIdx.y * p.src_stride;      // batch of independent transforms
    p.dst += (size_t)blockIdx.y * p.dst_stride;
    const uint32_t logR = p.logR, logS = p.logS, logC = p.logC;
    const uint32_t R = 1u << logR, C = 1u << logC, pitch = C + 1u;
    uint32_t* tile = smem;              // R * pitch words
    uint32_t* twl = smem + R * pitch;   // R/2 twiddles w_R^j
    const uint32_t tid = threadIdx.x;
    const uint32_t col0 = blockIdx.x << logC;
    const uint32_t tile_elems = R << logC;
    const uint32_t tw_shift = p.L - logR - logS;   // w_{RS} = root^(1 << tw_shift)
    const uint32_t smask = (1u << logS) - 1u;

    for (uint32_t j = tid; j < (R >> 1); j += kNttThreads) twl[j] = pow_lookup(p.tw, j << (p.L - logR));

    // ---- load ---------------------------------------------------------------
    if (MODE == NTT_DIT_LDE) {
        // S = B.  Each (prepared) coefficient feeds B columns: the innermost radix-B stage of the
        // size-N transform sees (c, 0, ..., 0) and is a replication.
        const uint32_t logB = logS;
        const uint32_t a0 = col0 >> logB;
        for (uint32_t u = tid; u < (tile_elems >> logB); u += kNttThreads) {
            uint32_t ao = u >> logR, t = u & (R - 1u);
            uint32_t v = p.src[((size_t)(a0 + ao) << logR) | t];
            uint32_t row = __brev(t) >> (32u - logR);
            uint32_t* dstp = &tile[row * pitch + (ao << logB)];
            dstp[0] = v;
            for (uint32_t s = 1; s <= smask; ++s) dstp[s] = mont_mul(v, pow_lookup(p.tw, (t * s) << tw_shift));
        }
    } else {
        for (uint32_t l = tid; l < tile_elems; l += kNttThreads) {
            uint32_t t, c;
            tile_coords(l, logR, logS, logC, t, c);
            uint32_t col = col0 + c, a = col >> logS, s = col & smask;
            size_t addr = ((size_t)a << (logR + logS)) | ((size_t)t << logS) | s;
            uint32_t v = p.src[addr];
            uint32_t row = t;
            if (MODE == NTT_DIT) {
                if (logS) v = mont_mul(v, pow_lookup(p.tw, (t * s) << tw_shift));
                row = __brev(t) >> (32u - logR);
            }
            tile[row * pitch + c] = v;
        }
    }
    __syncthreads();

    // ---- log2(R) radix-2 stages in LDS ------------------------------------------
    const uint32_t nb = tile_elems >> 1;
    if (MODE == NTT_DIF) {   // Gentleman-Sande: natural in, bit-reversed out
        for (int ll = (int)logR - 1; ll >= 0; --ll) {
            for (uint32_t b = tid; b < nb; b += kNttThreads) {
                uint32_t c = b & (C - 1u), pi = b >> logC;
                uint32_t j = pi & ((1u << ll) - 1u);
                uint32_t i = ((pi >> ll) << (ll + 1)) | j;
                uint32_t* x0 = &tile[i * pitch + c];
                uint32_t* x1 = x0 + (pitch << ll);
                uint32_t u = *x0, v = *x1;
                *x0 = add(u, v);
                *x1 = mont_mul(sub(u, v), twl[j << (logR - 1u - ll)]);
            }
            __syncthreads();
        }
    } else {                 // Cooley-Tukey: bit-reversed in, natural out
        for (uint32_t ll = 0; ll < logR; ++ll) {
            for (uint32_t b = tid; b < nb; b += kNttThreads) {
                uint32_t c = b & (C - 1u), pi = b >> logC;
                uint32_t j = pi & ((1u << ll) - 1u);
                uint32_t i = ((pi >> ll) << (ll + 1)) | j;
                uint32_t* x0 = &tile[i * pitch + c];
                uint32_t* x1 = x0 + (pitch << ll);
                uint32_t u = *x0, v = mont_mul(*x1, twl[j << (logR - 1u - ll)]);
                *x0 = add(u, v);
                *x1 = sub(u, v);
            }
            __syncthreads();
        }
    }

    // ---- store --------------------------------------------------------------
    for (uint32_t l = tid; l < tile_elems; l += kNttThreads) {
        uint32_t t, c;
        tile_coords(l, logR, logS, logC, t, c);
        uint32_t col = col0 + c, a = col >> logS, s = col & smask;
        size_t addr = ((size_t)a << (logR + logS)) | ((size_t)t << logS) | s;
        uint32_t v;
        if (MODE == NTT_DIF) {
            v = tile[(__brev(t) >> (32u - logR)) * pitch + c];
            if (logS) v = mont_mul(v, pow_lookup(p.tw, (t * s) << tw_shift));
            else if (p.scale_mont) v = mont_mul(v, p.scale_mont);
        } else {
            v = tile[t * pitch + c];
        }
        p.dst[addr] = v;
    }
}

bool launch_ntt_pass_fast(const NttPassArgs& a, NttMode mode, hipStream_t s, hipError_t* err);   // ntt_fast.hip

hipError_t launch_ntt_pass(const NttPassArgs& a, NttMode mode, hipStream_t s, Profiler* prof) {
    // algorithmic bytes: every element read once and written once (the LDE pass reads n, writes N)
    double bytes = mode == NTT_DIT_LDE ? 4.0 * ((double)((size_t)1 << (a.log_total - a.logS)) + (double)((size_t)1 << a.log_total))
                                       : 8.0 * (double)((size_t)1 << a.log_total);
    const uint32_t batch = a.batch ? a.batch : 1;
    ScopedKernelTimer tm(prof, K_NTT, bytes * batch, s, kNttOpsPerElement * (double)((size_t)1 << a.log_total) * batch * (double)a.logR / 7.0);
    hipError_t ferr = hipSuccess;
    if (launch_ntt_pass_fast(a, mode, s, &ferr)) return ferr;
    uint32_t cols_log = a.log_total - a.logR;
    uint32_t blocks = 1u << (cols_log - a.logC);
    size_t shmem = ((size_t)(1u << a.logR) * ((1u << a.logC) + 1u) + (1u << a.logR) / 2u) * sizeof(uint32_t);
    switch (mode) {
        case NTT_DIF: hipLaunchKernelGGL(ntt_pass_kernel<NTT_DIF>, dim3(blocks, batch), dim3(kNttThreads), shmem, s, a); break;
        case NTT_DIT: hipLaunchKernelGGL(ntt_pass_kernel<NTT_DIT>, dim3(blocks, batch), dim3(kNttThreads), shmem, s, a); break;
        case NTT_DIT_LDE: hipLaunchKernelGGL(ntt_pass_kernel<NTT_DIT_LDE>, dim3(blocks, batch), dim3(kNttThreads), shmem, s, a); break;
    }
    return hipGetLastError();
}

// Interpolant coefficients for the LDE, from the unscaled DIF output U of (a_0..a_{n-2}, 0):
//   out[pos] = (U[pos] - U[n-1] g^(k+1)) * shift^k / n,   k = true index of storage position pos.
// The first term removes the degree-(n-1) coefficient (the reference interpolates n-1 points,
// prover.rs:60: "virtual last trace point"), shift^k moves to the coset (prover.rs:69).
__global__ __launch_bounds__(256) void coef_prepare_kernel(const uint32_t* U, uint32_t* out, CoefPrepArgs a, size_t u_stride, size_t out_stride) {
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << a.log_n;
    if (pos >= n) return;
    U += (size_t)blockIdx.y * u_stride;
    out += (size_t)blockIdx.y * out_stride;
    uint32_t k = 0, rem = a.log_n, sh = 0;
    for (uint32_t d = 0; d < a.nd; ++d) {
        rem -= a.dig_bits[d];
        k |= ((pos >> rem) & ((1u << a.dig_bits[d]) - 1u)) << sh;
        sh += a.dig_bits[d];
    }
    const uint32_t c_top = U[n - 1u];
    uint32_t v = sub(U[pos], mont_mul(c_top, pow_lookup(a.tw, ((k + 1u) & (n - 1u)) << a.log_b)));
    out[pos] = mont_mul(v, mont_mul(pow_lookup(a.wtab, k), a.ninv_mont));
}

hipError_t launch_coef_prepare(const uint32_t* U, uint32_t* out, const CoefPrepArgs& a, hipStream_t s, Profiler* prof,
                               uint32_t batch, size_t u_stride, size_t out_stride) {
    uint32_t n = 1u << a.log_n;
    ScopedKernelTimer tm(prof, K_NTT, 8.0 * n * batch, s);
    hipLaunchKernelGGL(coef_prepare_kernel, dim3((n + 255) / 256, batch), dim3(256), 0, s, U, out, a, u_stride, out_stride);
    return hipGetLastError();
}

struct DigitArgs {
    uint32_t nd;
    uint32_t bits[kMaxDigits];
};

// to_natural: out[true_index(pos)] = in[pos]; else out[pos] = in[true_index(pos)]
__global__ void digit_reverse_kernel(const uint32_t* in, uint32_t* out, uint32_t log_m, DigitArgs dg, int to_natural) {
    size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= ((size_t)1 << log_m)) return;
    uint32_t k = 0, rem = log_m, sh = 0;
    for (uint32_t d = 0; d < dg.nd; ++d) {
        rem -= dg.bits[d];
        k |= (((uint32_t)pos >> rem) & ((1u << dg.bits[d]) - 1u)) << sh;
        sh += dg.bits[d];
    }
    if (to_natural) out[k] = in[pos];
    else out[pos] = in[k];
}

hipError_t launch_digit_reverse(const uint32_t* in, uint32_t* out, uint32_t log_m, uint32_t nd,
                                const uint32_t* dig_bits, int to_natural, hipStream_t s) {
    DigitArgs dg{};
    dg.nd = nd;
    for (uint32_t i = 0; i < nd; ++i) dg.bits[i] = dig_bits[i];
    size_t m = (size_t)1 << log_m;
    uint32_t threads = 256, blocks = (uint32_t)((m + threads - 1) / threads);
    hipLaunchKernelGGL(digit_reverse_kernel, dim3(blocks), dim3(threads), 0, s, in, out, log_m, dg, to_natural);
    return hipGetLastError();
}

// Degree check of the opt-in reference self-checks (prover.rs:154-156, :169, :228-251): coef is the unscaled DIF output
// (mixed-radix digit-reversed order) of a layer; out[0] += number of non-zero coefficients of true index >= bound,
// out[1] = the coefficient of true index bound - 1 (the asserted degree must be reached exactly).
__global__ void degree_check_kernel(const uint32_t* coef, uint32_t log_m, DigitArgs dg, uint32_t bound, uint32_t* out) {
    size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= ((size_t)1 << log_m)) return;
    uint32_t k = 0, rem = log_m, sh = 0;
    for (uint32_t d = 0; d < dg.nd; ++d) {
        rem -= dg.bits[d];
        k |= (((uint32_t)pos >> rem) & ((1u << dg.bits[d]) - 1u)) << sh;
        sh += dg.bits[d];
    }
    const uint32_t v = coef[pos];
    if (k >= bound && v != 0) atomicAdd(&out[0], 1u);
    if (k + 1 == bound) out[1] = v;
}
hipError_t launch_degree_check(const uint32_t* coef, uint32_t log_m, uint32_t nd, const uint32_t* dig_bits, uint32_t bound, uint32_t* out, hipStream_t s) {
    DigitArgs dg{};
    dg.nd = nd;
    for (uint32_t i = 0; i < nd; ++i) dg.bits[i] = dig_bits[i];
    size_t m = (size_t)1 << log_m;
    hipLaunchKernelGGL(degree_check_kernel, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, s, coef, log_m, dg, bound, out);
    return hipGetLastError();
}

// ===========================================================================
// Composition
// ===========================================================================

// Domain setup: inv_xm1[i] = 1/(shift h^i - 1), Montgomery form.  Each thread inverts
// kInvBatch denominators with one Fermat inversion (Montgomery's trick).
constexpr int kInvBatch = 8;

__global__ __launch_bounds__(256) void build_inv_xm1_kernel(uint32_t* out, size_t count, uint32_t e0, uint32_t order_mask, PowTable htab, uint32_t w_mont) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t base = gid * kInvBatch;
    if (base >= count) return;
    uint32_t d[kInvBatch], pre[kInvBatch];
    uint32_t acc = R1;
#pragma unroll
    for (int e = 0; e < kInvBatch; ++e) {
        size_t i = base + e;
        uint32_t x = i < count ? mont_mul(pow_lookup(htab, (e0 + (uint32_t)i) & order_mask), w_mont) : add(R1, R1);
        d[e] = sub(x, R1);
        pre[e] = acc;
        acc = mont_mul(acc, d[e]);
    }
    uint32_t inv = mont_inv(acc);
#pragma unroll
    for (int e = kInvBatch - 1; e >= 0; --e) {
        size_t i = base + e;
        uint32_t r = mont_mul(inv, pre[e]);
        inv = mont_mul(inv, d[e]);
        if (i < count) out[i] = r;
    }
}

hipError_t launch_build_inv_xm1_range(uint32_t* out, size_t count, uint32_t e0, uint32_t log_order, PowTable htab, uint32_t shift_mont, hipStream_t s) {
    size_t threads_total = (count + kInvBatch - 1) / kInvBatch;
    uint32_t blocks = (uint32_t)((threads_total + 255) / 256);
    if (!blocks) return hipSuccess;
    hipLaunchKernelGGL(build_inv_xm1_kernel, dim3(blocks), dim3(256), 0, s, out, count, e0, (uint32_t)(((uint64_t)1 << log_order) - 1u), htab, shift_mont);
    return hipGetLastError();
}
hipError_t launch_build_inv_xm1(uint32_t* out, uint32_t logN, PowTable htab, uint32_t shift_mont, hipStream_t s) {
    return launch_build_inv_xm1_range(out, (size_t)1 << logN, 0u, logN, htab, shift_mont, s);
}

// cp(x_i) for x_i = w h^i (prover.rs:101-166 evaluated pointwise; the same formula the
// verifier recomputes at proof.rs:63-77):
//   p0 = (f(x) - a[0]) / (x - 1)
//   p1 = (f(x) - a[n-2]) / (x - g^(n-2))            = (f(x) - a[n-2]) * g^2 / (x_{i+2B} - 1)
//   p2 = (f(g^2 x) - f(g x)^2 - f(x)^2) (x - g^(n-3))(x - g^(n-2))(x - g^(n-1)) / (x^n - 1)
//   cp = alpha0 p0 + alpha1 p1 + alpha2 p2
// with f(g x_i) = f[i+B], f(g^2 x_i) = f[i+2B] (indices mod N); x^n - 1 depends on i mod B only.
// BATCH: f and the trace / challenge constants belong to one proof of a batch, and a.zz carries no alpha2.
struct ComposeRaw { uint32_t f0, f1, f2, inv0, inv2, x_hi, x_lo, zz; };
__device__ __forceinline__ ComposeRaw compose_fetch(const ComposeArgs& a, const uint32_t* f, size_t i) {
    const size_t N = (size_t)1 << a.logN;
    const uint32_t B = 1u << a.log_b;
    const size_t i1 = (i + B) & (N - 1), i2 = (i + 2 * (size_t)B) & (N - 1);
    ComposeRaw r;
    r.f0 = f[i]; r.f1 = f[i1]; r.f2 = f[i2];
    r.inv0 = a.inv_xm1[i]; r.inv2 = a.inv_xm1[i2];
    r.x_hi = a.htab.hi[(uint32_t)i >> a.htab.lo_bits]; r.x_lo = a.htab.lo[(uint32_t)i & ((1u << a.htab.lo_bits) - 1u)];   // pow_lookup's two halves
    r.zz = a.zz[i & (B - 1)];
    return r;
}
// one value of cp from its operands: f at the three taps, 1/(x - 1) at two, x itself (Montgomery), zz = alpha2 / (x^n - 1) R^2
template <bool BATCH>
__device__ __forceinline__ uint32_t compose_eval(const ComposeArgs& a, uint32_t f0, uint32_t f1, uint32_t f2, uint32_t inv0, uint32_t inv2,
                                                 uint32_t x, uint32_t zz, uint32_t first, uint32_t last, uint32_t al0, uint32_t al1g2, uint32_t al2) {
    uint32_t t0 = mont_mul(mont_mul(sub(f0, first), inv0), al0);
    uint32_t t1 = mont_mul(mont_mul(sub(f0, last), inv2), al1g2);
    uint32_t v3 = mont_mul(mont_mul(sub(x, a.gm3_mont), sub(x, a.gm2_mont)), sub(x, a.gm1_mont));   // V*R
    uint32_t y = mont_mul(v3, zz);                                               // alpha2 V / (x^n-1) * R^2
    if (BATCH) y = mont_mul(y, al2);
    // data*data products carry R^-1; bring f2 to the same scale, fix with the R^2 in y
    uint32_t num = sub(sub(mont_mul(f2, 1u), mont_mul(f1, f1)), mont_mul(f0, f0));
    uint32_t t2 = mont_mul(num, y);
    return add(add(t0, t1), t2);
}
template <bool BATCH>
__device__ __forceinline__ uint32_t compose_finish(const ComposeArgs& a, const ComposeRaw& r, uint32_t first, uint32_t last, uint32_t al0,
                                                   uint32_t al1g2, uint32_t al2) {
    const uint32_t x = mont_mul(mont_mul(r.x_hi, r.x_lo), a.w_mont);              // Montgomery x_i
    return compose_eval<BATCH>(a, r.f0, r.f1, r.f2, r.inv0, r.inv2, x, r.zz, first, last, al0, al1g2, al2);
}
template <bool BATCH>
__device__ __forceinline__ uint32_t compose_core(const ComposeArgs& a, const uint32_t* f, uint32_t first, uint32_t last, uint32_t al0,
                                                 uint32_t al1g2, uint32_t al2, size_t i) {
    return compose_finish<BATCH>(a, compose_fetch(a, f, i), first, last, al0, al1g2, al2);
}
__device__ __forceinline__ uint32_t compose_at(const ComposeArgs& a, size_t i) {
    return compose_core<false>(a, a.f, a.first, a.last, a.alpha0_mont, a.alpha1g2_mont, 0u, i);
}

__global__ __launch_bounds__(256) void compose_kernel(ComposeArgs a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << a.logN)) return;
    a.cp[i] = compose_at(a, i);
}

// Four consecutive values per thread, 16-byte accesses (the stand-alone stage: HBM-bound, and one value per thread leaves
// the memory system short of requests).  The taps i + B and i + 2B are whole vectors when B >= 4; for B = 1, 2 (the local
// domains of a proof sharded over 8 or 4 GPUs) they are windows of the two vectors at i and i + 4.  x advances by h.
__global__ __launch_bounds__(256) void compose_kernel4(ComposeArgs a) {
    const size_t N = (size_t)1 << a.logN;
    const uint32_t B = 1u << a.log_b;
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= N) return;
    const uint4* fv = reinterpret_cast<const uint4*>(a.f);
    const uint4* iv = reinterpret_cast<const uint4*>(a.inv_xm1);
    uint32_t f0[4], f1[4], f2[4], inv0[4], inv2[4], zz[4];
    const uint4 v0 = fv[i >> 2], w0 = iv[i >> 2];
    f0[0] = v0.x; f0[1] = v0.y; f0[2] = v0.z; f0[3] = v0.w;
    inv0[0] = w0.x; inv0[1] = w0.y; inv0[2] = w0.z; inv0[3] = w0.w;
    if (B >= 4) {
        const uint4 v1 = fv[((i + B) & (N - 1)) >> 2], v2 = fv[((i + 2 * (size_t)B) & (N - 1)) >> 2], w2 = iv[((i + 2 * (size_t)B) & (N - 1)) >> 2];
        f1[0] = v1.x; f1[1] = v1.y; f1[2] = v1.z; f1[3] = v1.w;
        f2[0] = v2.x; f2[1] = v2.y; f2[2] = v2.z; f2[3] = v2.w;
        inv2[0] = w2.x; inv2[1] = w2.y; inv2[2] = w2.z; inv2[3] = w2.w;
        const uint32_t z0 = (uint32_t)i & (B - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) zz[k] = a.zz[z0 + k];
    } else {
        const uint4 v1 = fv[((i + 4) & (N - 1)) >> 2], w1 = iv[((i + 4) & (N - 1)) >> 2];
        const uint32_t fw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const uint32_t iw[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f1[k] = B == 1 ? fw[k + 1] : fw[k + 2];
            f2[k] = B == 1 ? fw[k + 2] : fw[k + 4];
            inv2[k] = B == 1 ? iw[k + 2] : iw[k + 4];
            zz[k] = B == 1 ? a.zz[0] : a.zz[k & 1];
        }
    }
    uint32_t x = mont_mul(pow_lookup(a.htab, (uint32_t)i), a.w_mont);
    const uint32_t h = pow_lookup(a.htab, 1u);
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = compose_eval<false>(a, f0[k], f1[k], f2[k], inv0[k], inv2[k], x, zz[k], a.first, a.last, a.alpha0_mont, a.alpha1g2_mont, 0u);
        if (k < 3) x = mont_mul(x, h);
    }
    reinterpret_cast<uint4*>(a.cp)[i >> 2] = make_uint4(r[0], r[1], r[2], r[3]);
}

hipError_t launch_compose(const ComposeArgs& a, hipStream_t s, Profiler* prof) {
    size_t N = (size_t)1 << a.logN;
    ScopedKernelTimer tm(prof, K_COMPOSE, 8.0 * (double)N, s);   // read f once, write cp once
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.f) | reinterpret_cast<uintptr_t>(a.inv_xm1) | reinterpret_cast<uintptr_t>(a.cp)) & 15u) == 0;
    if (a.logN >= 4 && aligned) {
        uint32_t blocks = (uint32_t)((N / 4 + 255) / 256);
        hipLaunchKernelGGL(compose_kernel4, dim3(blocks), dim3(256), 0, s, a);
    } else {
        uint32_t blocks = (uint32_t)((N + 255) / 256);
        hipLaunchKernelGGL(compose_kernel, dim3(blocks), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

// ===========================================================================
// FRI fold
// ===========================================================================
// next[i] = (e[i] + e[i+m/2])/2 + beta (e[i] - e[i+m/2]) / (2 x_i),  x_i = (w h^i)^(2^r)
// (identity pinned by fri_test, polynomial.rs:418-425, and used at proof.rs:110-113).
struct FoldRaw { uint32_t u, v, x_hi, x_lo; };
__device__ __forceinline__ FoldRaw fold_fetch(const FoldArgs& a, size_t i) {
    const size_t half = (size_t)1 << (a.log_m - 1);
    const uint32_t e = (uint32_t)(i << a.round);                    // h^(-2^r i), i 2^r < N/2
    FoldRaw r;
    r.u = a.in[i]; r.v = a.in[i + half];
    r.x_hi = a.hinv.hi[e >> a.hinv.lo_bits]; r.x_lo = a.hinv.lo[e & ((1u << a.hinv.lo_bits) - 1u)];
    return r;
}
__device__ __forceinline__ uint32_t fold_finish(const FoldArgs& a, const FoldRaw& r) {
    uint32_t xinv = mont_mul(r.x_hi, r.x_lo);
    uint32_t s = mont_mul(add(r.u, r.v), a.inv2_mont);
    uint32_t d = mont_mul(mont_mul(sub(r.u, r.v), xinv), a.c_mont);
    return add(s, d);
}
__device__ __forceinline__ uint32_t fold_at(const FoldArgs& a, size_t i) { return fold_finish(a, fold_fetch(a, i)); }

__global__ __launch_bounds__(256) void fri_fold_kernel(FoldArgs a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << (a.log_m - 1))) return;
    a.out[i] = fold_at(a, i);
}

// four consecutive outputs per thread, 16-byte accesses; x^-1 advances by h^(-2^r)
__global__ __launch_bounds__(256) void fri_fold_kernel4(FoldArgs a) {
    const size_t half = (size_t)1 << (a.log_m - 1);
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= half) return;
    const uint4 u = reinterpret_cast<const uint4*>(a.in)[i >> 2], v = reinterpret_cast<const uint4*>(a.in)[(i + half) >> 2];
    const uint32_t uu[4] = {u.x, u.y, u.z, u.w}, vv[4] = {v.x, v.y, v.z, v.w};
    uint32_t xinv = pow_lookup(a.hinv, (uint32_t)(i << a.round));
    const uint32_t step = pow_lookup(a.hinv, 1u << a.round);
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sm = mont_mul(add(uu[k], vv[k]), a.inv2_mont);
        const uint32_t d = mont_mul(mont_mul(sub(uu[k], vv[k]), xinv), a.c_mont);
        r[k] = add(sm, d);
        if (k < 3) xinv = mont_mul(xinv, step);
    }
    reinterpret_cast<uint4*>(a.out)[i >> 2] = make_uint4(r[0], r[1], r[2], r[3]);
}

hipError_t launch_fri_fold(const FoldArgs& a, hipStream_t s, Profiler* prof) {
    size_t half = (size_t)1 << (a.log_m - 1);
    ScopedKernelTimer tm(prof, K_FOLD, 12.0 * (double)half, s);   // read m words, write m/2
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out)) & 15u) == 0;
    if (half >= 4 && aligned) {
        uint32_t blocks = (uint32_t)((half / 4 + 255) / 256);
        hipLaunchKernelGGL(fri_fold_kernel4, dim3(blocks), dim3(256), 0, s, a);
    } else {
        uint32_t blocks = (uint32_t)((half + 255) / 256);
        hipLaunchKernelGGL(fri_fold_kernel, dim3(blocks), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

// ===========================================================================
// Merkle tree
// ===========================================================================
//
// Heap layout of merkle.rs:14-51: 2m-1 nodes, root 0, children of j at 2j+1, 2j+2,
// depth d occupies [2^d - 1, 2^(d+1) - 1), leaf i at m - 1 + i.
//
// SHA-256 is integer-VALU bound, so the design goal is 100 % lane utilisation with every
// HBM access a full line.  A wave owns 64 * 2^k consecutive inputs and walks them in groups of
// 64: group i is hashed by the 64 lanes (one coalesced 256 B / 2 KiB access), two sibling groups
// are paired through a 4 KiB LDS buffer into the 64 parents (lane j hashes children 2j, 2j+1), and
// so on up k levels like a binary counter -- every level keeps all 64 lanes busy and every
// digest is written once in a contiguous 2 KiB run.  All control flow is wave-uniform.  One launch lowers the tree by k levels; the last
// <= 2^11 nodes are finished by a single workgroup that keeps the level in LDS.

// Merkle hash selector: 0 = SHA-256 (merkle.rs:1-2), 1 = field-native hash (fieldhash.hpp).
__constant__ FieldHashConsts g_fh_consts;       // Montgomery form: the 16-lane row form of the narrow levels (fieldhash_inner_row16)
__constant__ FieldHashConsts64 g_fh_consts64;   // canonical residues as doubles: every other field hash on the device (fieldhash_f64.hpp)

template <int HASH> struct Hasher;
template <> struct Hasher<0> {
    static __device__ __forceinline__ Digest leaf(uint32_t v) { return sha256_leaf(v); }
    static __device__ __forceinline__ Digest inner(const Digest& l, const Digest& r) { return sha256_inner(l, r); }
};
// The field hash runs in double precision on the device (round 5): the same function bit for bit, 0.74 of the time of the
// 32-bit Montgomery form in a dependent chain (tools/fh64_probe.hip, profiles/r05_fh64_probe.txt): an addition is one
// instruction instead of four when nothing has to be corrected.
template <> struct Hasher<1> {
    static __device__ __forceinline__ Digest leaf(uint32_t v) { return fieldhash_leaf64(v, g_fh_consts64); }
    static __device__ __forceinline__ Digest inner(const Digest& l, const Digest& r) { return fieldhash_inner64(l, r, g_fh_consts64); }
};

// Where the leaf values of a tree come from.  The prover fuses the elementwise producer of a layer
// into the leaf hashing of its commitment: the value is computed, written to the layer and hashed
// in one pass (no separate fold / compose launch, the layer is never re-read for hashing).
// A source is read in two steps so that the throughput kernel can have the memory reads of its NEXT group of 64 leaves
// in flight while it hashes the current one: fetch() only issues loads (what they return is `Raw`), finish() does the
// arithmetic and the store of the produced value.  load() = both, for the callers that take one value at a time.
struct PlainSrc {
    const uint32_t* vals;
    using Raw = uint32_t;
    __device__ __forceinline__ Raw fetch(size_t pos) const { return vals[pos]; }
    __device__ __forceinline__ uint32_t finish(const Raw& r, size_t) const { return r; }
    __device__ __forceinline__ uint32_t load(size_t pos) const { return vals[pos]; }
};
struct FoldSrc {       // FRI layer r+1 = fold(layer r, beta): prover.rs:198-211 + :214
    FoldArgs a;
    using Raw = FoldRaw;
    __device__ __forceinline__ void prepare() { if (a.dyn) a.c_mont = *a.dyn; }   // early launch: the challenge arrived after the enqueue
    __device__ __forceinline__ Raw fetch(size_t pos) const { return fold_fetch(a, pos); }
    __device__ __forceinline__ uint32_t finish(const Raw& r, size_t pos) const { uint32_t v = fold_finish(a, r); a.out[pos] = v; return v; }
    __device__ __forceinline__ uint32_t load(size_t pos) const { return finish(fetch(pos), pos); }
};
struct ComposeSrc {    // cp layer 0 from f_eval: prover.rs:101-173 + :176
    ComposeArgs a;
    using Raw = size_t;        // nothing fetched ahead: eight words in flight per lane cost more registers than the 2^24-leaf
                               // launch (four waves per SIMD hide the latency) gains -- measured 1 579 against 1 560 us
    __device__ __forceinline__ Raw fetch(size_t pos) const { return pos; }
    __device__ __forceinline__ uint32_t finish(const Raw& r, size_t) const { return load(r); }
    __device__ __forceinline__ uint32_t load(size_t pos) const { uint32_t v = compose_at(a, pos); a.cp[pos] = v; return v; }
};

struct ComposeBlockSrc {   // cp over this rank's block of a sharded proof, from the block of f it received (ComposeBlockArgs)
    ComposeBlockArgs b;
    using Raw = size_t;
    __device__ __forceinline__ uint32_t f_at(size_t t) const {
        const uint32_t gmask = (1u << b.lg) - 1u;
        if (t >> b.log_m) {                                     // the 2B positions after the block
            const size_t v = t - ((size_t)1 << b.log_m);
            return b.halo[(v & gmask) * b.halo_stride + (v >> b.lg)];
        }
        const uint32_t log_cl = b.lg + b.log_cnt;               // leaves per chunk of the exchange
        const size_t piece = ((t >> log_cl) << b.lg) | (t & gmask);
        return b.a.f[(piece << b.log_cnt) | ((t & (((size_t)1 << log_cl) - 1)) >> b.lg)];
    }
    __device__ __forceinline__ Raw fetch(size_t pos) const { return pos; }
    __device__ __forceinline__ uint32_t finish(const Raw& r, size_t) const { return load(r); }
    __device__ __forceinline__ uint32_t load(size_t pos) const {
        const ComposeArgs& a = b.a;
        const size_t B = (size_t)1 << a.log_b;
        const uint32_t f0 = f_at(pos), f1 = f_at(pos + B), f2 = f_at(pos + 2 * B);
        const uint32_t x = mont_mul(pow_lookup(a.htab, b.e0 + (uint32_t)pos), a.w_mont);
        return compose_eval<false>(a, f0, f1, f2, a.inv_xm1[pos], a.inv_xm1[pos + 2 * B], x, a.zz[pos & (B - 1)], a.first, a.last, a.alpha0_mont,
                                   a.alpha1g2_mont, 0u);
    }
};

struct ComposeBatchSrc {   // batch of proofs: leaf b*N + i = cp_b[i], with proof b's own challenges
    ComposeArgs a;
    const BatchChal* chal;
    using Raw = size_t;        // batched trees run with the chip full: nothing fetched ahead
    __device__ __forceinline__ Raw fetch(size_t pos) const { return pos; }
    __device__ __forceinline__ uint32_t finish(const Raw& r, size_t) const { return load(r); }
    __device__ __forceinline__ uint32_t load(size_t pos) const {
        const size_t b = pos >> a.logN, i = pos & (((size_t)1 << a.logN) - 1);
        const BatchChal c = chal[b];
        uint32_t v = compose_core<true>(a, a.f + (b << a.logN), c.first, c.last, c.alpha0_mont, c.alpha1g2_mont, c.alpha2_mont, i);
        a.cp[pos] = v;
        return v;
    }
};
struct FoldBatchSrc {      // leaf b*(m/2) + i = fold of proof b's layer with its own beta
    FoldArgs a;
    const BatchChal* chal;
    using Raw = size_t;
    __device__ __forceinline__ Raw fetch(size_t pos) const { return pos; }
    __device__ __forceinline__ uint32_t finish(const Raw& r, size_t) const { return load(r); }
    __device__ __forceinline__ uint32_t load(size_t pos) const {
        const uint32_t lh = a.log_m - 1;
        const size_t half = (size_t)1 << lh, b = pos >> lh, i = pos & (half - 1);
        const uint32_t* in = a.in + (b << a.log_m);
        uint32_t u = in[i], v = in[i + half];
        uint32_t xinv = pow_lookup(a.hinv, (uint32_t)(i << a.round));
        uint32_t s = mont_mul(add(u, v), a.inv2_mont);
        uint32_t d = mont_mul(mont_mul(sub(u, v), xinv), chal[b].c_mont);
        uint32_t r = add(s, d);
        a.out[pos] = r;
        return r;
    }
};

struct InterleaveSrc { // leaves arrive as 2^log_parts cyclic pieces of 2^log_cnt words (multi-GPU all-to-all output):
    const uint32_t* recv;  // leaf u*parts + q = recv[q*cnt + u]; hashed straight from the receive buffer
    uint32_t log_parts, log_cnt;
    using Raw = uint32_t;
    __device__ __forceinline__ uint32_t load(size_t pos) const {
        return recv[((pos & (((size_t)1 << log_parts) - 1)) << log_cnt) | (pos >> log_parts)];
    }
    __device__ __forceinline__ Raw fetch(size_t pos) const { return load(pos); }
    __device__ __forceinline__ uint32_t finish(const Raw& r, size_t) const { return r; }
};

// Sources whose constants may arrive after the launch was enqueued have a prepare(); every kernel calls it on its own copy of the
// source before the first element is produced.
template <class S> __device__ __forceinline__ auto src_prepare(S& s, int) -> decltype(s.prepare(), void()) { s.prepare(); }
template <class S> __device__ __forceinline__ void src_prepare(S&, long) {}

constexpr int kMerkleThreads = 256;
constexpr uint32_t kMerkleMaxK = 4;

__device__ __forceinline__ void store_digest(uint32_t* nodes, size_t node, const Digest& d) {
    uint4* q = reinterpret_cast<uint4*>(nodes + node * 8);
    q[0] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]);
    q[1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
}
__device__ __forceinline__ Digest load_digest(const uint32_t* nodes, size_t node) {
    const uint4* q = reinterpret_cast<const uint4*>(nodes + node * 8);
    uint4 lo = q[0], hi = q[1];
    Digest d;
    d.w[0] = lo.x; d.w[1] = lo.y; d.w[2] = lo.z; d.w[3] = lo.w;
    d.w[4] = hi.x; d.w[5] = hi.y; d.w[6] = hi.z; d.w[7] = hi.w;
    return d;
}

__device__ __forceinline__ Digest lds_digest(const uint4* p) {
    uint4 lo = p[0], hi = p[1];
    Digest d;
    d.w[0] = lo.x; d.w[1] = lo.y; d.w[2] = lo.z; d.w[3] = lo.w;
    d.w[4] = hi.x; d.w[5] = hi.y; d.w[6] = hi.z; d.w[7] = hi.w;
    return d;
}

template <class SRC, bool LEAF, int HASH>
__global__ __launch_bounds__(kMerkleThreads) void merkle_subtree_kernel(SRC src, uint32_t* nodes,
                                                                        uint32_t depth_in, uint32_t k, size_t off) {
    // per wave: k pending LEFT sibling groups (one per level) + one scratch group for the right sibling of the
    // pairing in progress, 64 digests x 2 uint4 each = (k + 1) x 2 KiB: 40 KiB per workgroup at k = 4, four waves per
    // SIMD.  (Round 3 tried keeping only level 0 in LDS and re-reading the pending groups of higher levels from the
    // heap, where every digest is written anyway: 4 KiB per wave, <= 56 VGPRs, eight waves per SIMD and any k -- and
    // measured 6.19 ms per 2^24 proof against 6.07 ms for this kernel, profiles/r03_ab_subtree_heap.txt: SHA-256 as
    // compiled does not issue faster with more resident waves, and the second inlined copy of the hash costs more.)
    extern __shared__ __attribute__((aligned(16))) uint4 stage[];
    src_prepare(src, 0);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t gwave = (size_t)blockIdx.x * (kMerkleThreads / 64) + wave;
    // first input of this wave; `off` = position of a chunk's first node at this depth (0 for a whole
    // tree): a chunk is an aligned sub-range of the leaves, built into its place in the heap
    const size_t base = (gwave << (6 + k)) + off;
    const size_t in_base = ((size_t)1 << depth_in) - 1;
    uint4* my = stage + (size_t)wave * (k + 1) * 128;
    uint4* scratch = my + (size_t)k * 128;
    // The reads of group i + 1 are issued before group i is hashed (a wave that meets its load latency at the top of
    // every group stalls 2^k times, and with two waves per SIMD running the same code in step, both at once), and
    // they are consumed -- SRC::finish: the producer's arithmetic and the store of its value -- at the END of group i,
    // behind that group's digest stores: the wait in front of finish() then leaves those stores in flight.
    typename SRC::Raw raw;
    Digest dnext;
    uint32_t val = 0;
    if (LEAF) val = src.load(base + lane - off);                 // the source is chunk-local
    else dnext = load_digest(nodes, in_base + base + lane);
#pragma unroll 1
    for (uint32_t i = 0; i < (1u << k); ++i) {
        Digest d;
        const size_t pos = base + (size_t)i * 64 + lane;         // 64 consecutive inputs: coalesced
        const bool more = i + 1 < (1u << k);                     // wave-uniform
        if (LEAF) {
            if (more) raw = src.fetch(pos + 64 - off);
            d = Hasher<HASH>::leaf(val);
            store_digest(nodes, in_base + pos, d);
        } else {
            d = dnext;
            if (more) dnext = load_digest(nodes, in_base + pos + 64);
        }
        uint32_t idx = i, lvl = 0;
#pragma unroll 1
        while (lvl < k) {                                        // wave-uniform
            uint4* left = my + lvl * 128;
            uint4* grp = (idx & 1u) ? scratch : left;
            grp[2 * lane] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]);
            grp[2 * lane + 1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
            if (!(idx & 1u)) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // children 2*lane, 2*lane+1 of the 128 buffered nodes: lanes 0-31 pair the left group, lanes 32-63 the right
            const uint4* x = (lane < 32u ? left : scratch - 128) + 4 * lane;
            Digest l = lds_digest(x), r = lds_digest(x + 2);
            __builtin_amdgcn_wave_barrier();
            d = Hasher<HASH>::inner(l, r);
            idx >>= 1;
            ++lvl;
            store_digest(nodes, (((size_t)1 << (depth_in - lvl)) - 1) + (base >> lvl) + (size_t)idx * 64 + lane, d);
        }
        if (LEAF && more) val = src.finish(raw, pos + 64 - off);
    }
}

// Latency-bound part of a tree.  A level with <= 65 536 nodes costs one hash latency however it
// is scheduled (one wave per SIMD already issues a VALU op every 4 cycles), so what matters here
// is the number of grid-wide synchronisations.  Each workgroup takes 2^j consecutive nodes of
// level `depth_in` (raw values in LEAF mode), keeps them in LDS and reduces them to ONE node with
// workgroup barriers only; a launch therefore lowers the tree by j levels.
constexpr int kWgThreads = 256;
#ifndef ZK_FIELD_ROW_MAX_NODES
#define ZK_FIELD_ROW_MAX_NODES 16       // round 6, re-swept with the quad form present (profiles/r06_ab_field_forms.txt): 32 nodes are one quad pass (5.3 us), not two row passes (7.2)
#endif
constexpr uint32_t kFieldRowMaxNodes = ZK_FIELD_ROW_MAX_NODES;   // levels of <= this many nodes per workgroup use the 16-lane row form of the field hash
static_assert(kFieldRowMaxNodes >= 16 && kFieldRowMaxNodes <= 64, "the row form takes at most four passes of 16 nodes");
#ifndef ZK_FIELD_QUAD_MAX_NODES
#define ZK_FIELD_QUAD_MAX_NODES 64      // profiles/r05_ab_field_quad.txt: 0 / 64 / 128 swept; a quad pass is ~6 us, so two of them lose to the one-lane hash
#endif
constexpr uint32_t kFieldQuadMaxNodes = ZK_FIELD_QUAD_MAX_NODES;   // ... of <= this many (and more than the row form's) one hash per quad of lanes; 0: never
static_assert(kFieldQuadMaxNodes <= 128, "the quad form takes at most two passes of 64 nodes");
#ifndef ZK_FIELD_ROW_LEAF_MAX
#define ZK_FIELD_ROW_LEAF_MAX 32        // 0: leaves always one lane per hash
#endif
constexpr uint32_t kFieldRowLeafMax = ZK_FIELD_ROW_LEAF_MAX;     // a workgroup with <= this many LEAVES of the field hash hashes them in the row form too (round 6)
static_assert(kFieldRowLeafMax <= 64, "the row form takes at most four passes of 16 leaves");
constexpr uint32_t kWgMaxLog = 10;   // 1024 digests = 32 KiB LDS per workgroup

__device__ __forceinline__ void lds_store(uint4* p, const Digest& d) {
    p[0] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]);
    p[1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
}

// ---- handing a workgroup's result to the workgroup that finishes last (round 6) ----------------------------------------------
// The chip has eight compute dies with an L2 each.  Rounds 1-5 published a workgroup's node with an agent-scope RELEASE on the
// arrival counter (buffer_wbl2: write back the die's L2, ~0.5 MB of fresh digests per die in a 2^17-node launch) and the last
// workgroup took everything in with an ACQUIRE (buffer_inv).  tools/wg_trace.py put 5.4 - 8.3 us per latency launch between "a
// workgroup has its node" and "the last one starts the post" at EVERY tree size: that hand-over, not the hashing.  Only ONE node
// per workgroup has to cross: it is re-stored with eight write-through stores (relaxed atomic stores at agent scope: sc1), the
// wave waits for them, the counter is bumped RELAXED, and the last workgroup reads the nodes with agent-scope loads (sc1: they
// miss the die's L2).  Everything else a launch wrote becomes visible at the kernel boundary as always.
// Build-time A/B: ZK_BUILD_DEFS="-DZK_WG_RELAXED_PUBLISH=0" restores the release / acquire form.
#ifndef ZK_WG_RELAXED_PUBLISH
#define ZK_WG_RELAXED_PUBLISH 1
#endif
constexpr bool kWgRelaxedPublish = ZK_WG_RELAXED_PUBLISH != 0;
__device__ __forceinline__ void publish_node(uint32_t* nodes, size_t node, const uint4* lvl0, uint32_t tid) {
    if (tid < 8) __hip_atomic_store(nodes + node * 8 + tid, reinterpret_cast<const uint32_t*>(lvl0)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // wave 0: the eight stores are out before its thread 0 counts
}
__device__ __forceinline__ Digest load_digest_agent(const uint32_t* nodes, size_t node) {
    // four 8-byte agent-scope loads (global_load_dwordx2 ... sc1; a digest is 32-byte aligned)
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(nodes + node * 8);
    Digest d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned long long v = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d.w[2 * i] = (uint32_t)v; d.w[2 * i + 1] = (uint32_t)(v >> 32);
    }
    return d;
}

// ---- diagnostic build: where the time of a latency launch goes (ZK_BUILD_DEFS="-DZK_WG_TRACE=1"; tools/wg_trace.py) ----------
// Thread 0 of workgroup 0 stamps the 100 MHz constant clock (s_memrealtime: one time base for every compute unit) at entry,
// after the inputs are in LDS, and after every level; the workgroup that carries on (continuation) and the one that posts to the
// host stamp their steps into the same record.  Records are dumped to $ZK_WG_TRACE_FILE when a context is destroyed.  Not part
// of a product build: the stamps cost a scalar memory-time read and a store per level.
#ifdef ZK_WG_TRACE
constexpr uint32_t kWgTraceSlots = 48, kWgTraceLaunches = 8192;
__device__ unsigned long long g_wg_trace[kWgTraceLaunches][kWgTraceSlots];
__device__ unsigned int g_wg_trace_n, g_wg_trace_cur;
#define WG_STAMP(slot) do { if (threadIdx.x == 0 && wg_rec < kWgTraceLaunches && (slot) < kWgTraceSlots) g_wg_trace[wg_rec][(slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WG_STAMP(slot) do { } while (0)
#endif

// When the launch reaches the hand-over depth (MailArgs.top; 0 = the root) and a mailbox is given, the
// digests of that depth are also written to host-mapped memory followed by a sequence number, so the
// host prover can poll for them instead of paying a blit kernel + stream synchronisation per commitment.
// j2 > 0 (whole trees only, mail.counter given): the workgroup that finishes LAST goes on with the gridDim.x nodes the
// launch has produced (<= 2^kWgMaxLog of them) and lowers the tree by j2 more levels -- one launch where the levels of a
// tree used to need two (round 5: the field hash, which hands nothing to the host, paid 38 latency launches per proof).
template <class SRC, bool LEAF, int HASH>
__global__ __launch_bounds__(kWgThreads) void merkle_wg_kernel(SRC src, uint32_t* nodes, uint32_t depth_in, uint32_t j,
                                                               MailArgs mail, size_t off, uint32_t j2, uint32_t lds_log) {
    extern __shared__ __attribute__((aligned(16))) uint4 lvl[];   // [2^lds_log][2], then the 16 KiB schedule exchange
    src_prepare(src, 0);
    uint32_t* xch = reinterpret_cast<uint32_t*>(lvl + ((size_t)2 << lds_log));
    const uint32_t tid = threadIdx.x;
    uint32_t cnt = 1u << j;                                       // inputs of this workgroup in the current phase
    size_t first = ((size_t)blockIdx.x << j) + off;               // first input of this workgroup (off: see merkle_subtree_kernel)
    uint32_t d_in = depth_in, lv = j;                             // the current phase lowers depth d_in by lv levels
    const size_t in_base = ((size_t)1 << depth_in) - 1;
    QuadLane ql;
    if (HASH == 0) ql = quad_lane(tid);
#ifdef ZK_WG_TRACE
    uint32_t wg_rec = 0xffffffffu;                                // record of this launch; only workgroup 0 knows it until the hand-over
    if (blockIdx.x == 0 && tid == 0) {
        wg_rec = atomicAdd(&g_wg_trace_n, 1u);
        __hip_atomic_store(&g_wg_trace_cur, wg_rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wg_rec < kWgTraceLaunches) {
            for (uint32_t i = 0; i < kWgTraceSlots; ++i) g_wg_trace[wg_rec][i] = 0;
            g_wg_trace[wg_rec][0] = (unsigned long long)depth_in | ((unsigned long long)j << 8) | ((unsigned long long)j2 << 16) | ((unsigned long long)(LEAF ? 1 : 0) << 24) |
                                    ((unsigned long long)HASH << 25) | ((unsigned long long)(mail.mailbox ? mail.top + 1 : 0) << 26) | ((unsigned long long)gridDim.x << 32);
        }
    }
    WG_STAMP(1);
#endif
    // a thread takes up to 2^kWgMaxLog / kWgThreads = 4 inputs: every read is issued before the first one is used (one
    // load latency per launch instead of one per input; this phase is a chain of latencies)
    constexpr uint32_t kPer = (1u << kWgMaxLog) / kWgThreads;
    if (LEAF && HASH == 1 && cnt <= kFieldRowLeafMax) {
        // Few leaves of the field hash (the small trees of a proof's tail, spread over several workgroups): a leaf is the same
        // permutation as a node, on (v, 0, ..., 0, 1), so it takes the 16-lane row form as well -- 3.6 us per pass of 16 leaves where
        // the one-lane hash costs ~10.7 us however few there are (profiles/r06_wg_trace_21_field.txt: "load/leaf" of every leaf-mode
        // launch).  Every lane of a row asks the source for the row's leaf (one address: the loads coalesce; a fused producer computes
        // and stores the same value sixteen times over, which costs lanes that would idle anyway).
        const uint32_t g = tid & 15u, grp = tid >> 4;
#pragma unroll 1
        for (uint32_t b = 0; b * 16 < cnt; ++b) {
            const uint32_t i = b * 16 + grp;
            const uint32_t ii = i < cnt ? i : 0u;                  // idle rows recompute leaf 0 (DPP needs whole rows running)
            const uint32_t v = src.finish(src.fetch(first + ii - off), first + ii - off);
            const uint32_t word = g == 0 ? v : g == 15 ? 1u : 0u;
            const uint32_t res = fieldhash_inner_row16_f64(word, g, g_fh_consts64);
            if (i < cnt && g < 8) {
                nodes[(in_base + first + i) * 8 + g] = res;
                reinterpret_cast<uint32_t*>(&lvl[2 * i])[g] = res;
            }
        }
    } else if (LEAF) {
        typename SRC::Raw raw[kPer] = {};
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u)
            if (tid + u * kWgThreads < cnt) raw[u] = src.fetch(first + tid + u * kWgThreads - off);
#pragma unroll 1
        for (uint32_t i = tid; i < cnt; i += kWgThreads) {          // one copy of the hash: the fetched inputs move up
            const Digest d = Hasher<HASH>::leaf(src.finish(raw[0], first + i - off));
            store_digest(nodes, in_base + first + i, d);
            lds_store(&lvl[2 * i], d);
#pragma unroll
            for (uint32_t u = 0; u + 1 < kPer; ++u) raw[u] = raw[u + 1];
        }
    } else {
        Digest dd[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u)
            if (tid + u * kWgThreads < cnt) dd[u] = load_digest(nodes, in_base + first + tid + u * kWgThreads);
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u)
            if (tid + u * kWgThreads < cnt) lds_store(&lvl[2 * (tid + u * kWgThreads)], dd[u]);
    }
    // the layer values this workgroup produced are read by ANOTHER workgroup (the one that finishes last) when
    // they are posted to the host: make every wave's stores visible device-wide, not only the posting wave's
    if (LEAF && mail.dump_src) __threadfence();
    __syncthreads();
    WG_STAMP(2);
#pragma unroll 1
  for (uint32_t phase = 0;; ++phase) {
#pragma unroll 1
    for (uint32_t t = 1; t <= lv; ++t) {
        const uint32_t w = cnt >> t;                              // nodes of this level in the workgroup
        const size_t out_base = (((size_t)1 << (d_in - t)) - 1) + (first >> t);
        WG_STAMP((phase ? 24u : 2u) + t - 1u);                   // start of level t = end of level t - 1
        if (HASH == 0 && w <= 64) {
            // Latency-bound level of <= 64 nodes per workgroup: one SHA-256 per FOUR lanes (sha256_quad.hpp: ~1 700
            // instructions on the wave instead of 2 293, no exchange inside the hash), 16 hashes per wave, the waves of
            // the workgroup side by side on their own SIMDs.  Every lane of a wave that hashes runs the code (DPP).
            const uint32_t lane = tid & 63u, wave = tid >> 6;
            const uint32_t node = wave * 16 + (lane >> 4) * 4 + (lane & 3u), role = (lane >> 2) & 3u;
            const bool busy = wave * 16 < w;                          // wave-uniform
            uint32_t o[4];
            if (busy) sha256_inner_quad(reinterpret_cast<const uint32_t*>(&lvl[4 * (node < w ? node : 0u)]), ql, o);
            __syncthreads();                                       // every read of the level done
            if (busy && node < w && role < 2) {                    // bank 1 holds words 0-3, bank 0 words 4-7
                const uint4 v = make_uint4(o[0], o[1], o[2], o[3]);
                lvl[2 * node + (role ^ 1u)] = v;
                reinterpret_cast<uint4*>(nodes + (out_base + node) * 8)[role ^ 1u] = v;
            }
            __syncthreads();
            continue;
        }
        if (HASH == 0 && w <= 128) {
            // Latency-bound level with idle waves: split each SHA-256 between a main lane (waves 0-1:
            // the 2 x 64 rounds) and a helper lane (waves 2-3: the 48 message-schedule steps of block 1),
            // 16 rounds at a time through a double-buffered LDS exchange.  The main lane's critical
            // path drops from 2293 to ~1850 instructions.
            const bool is_main = tid < 128;
            const uint32_t node = tid & 127u;
            const bool active = node < w;
            uint32_t wv[16];
            uint32_t st[8];
            if (active) {
                Digest l = lds_digest(&lvl[4 * node]), r = lds_digest(&lvl[4 * node + 2]);
#pragma unroll
                for (int i = 0; i < 8; ++i) { wv[i] = l.w[i]; wv[8 + i] = r.w[i]; st[i] = SHA_IV[i]; }
            }
            uint32_t* xb0 = xch;                                   // [16][128]
            uint32_t* xb1 = xch + 16 * 128;
            // phase A: rounds 0-15 | W16-31 -> xb0
            if (active) {
                if (is_main) sha256_rounds16<0>(st, wv);
                else { sha256_schedule16(wv);
#pragma unroll
                       for (int i = 0; i < 16; ++i) xb0[i * 128 + node] = wv[i]; }
            }
            __syncthreads();
            // phase B: rounds 16-31 | W32-47 -> xb1
            if (active) {
                if (is_main) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) wv[i] = xb0[i * 128 + node];
                    sha256_rounds16<16>(st, wv);
                } else { sha256_schedule16(wv);
#pragma unroll
                         for (int i = 0; i < 16; ++i) xb1[i * 128 + node] = wv[i]; }
            }
            __syncthreads();
            // phase C: rounds 32-47 | W48-63 -> xb0
            if (active) {
                if (is_main) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) wv[i] = xb1[i * 128 + node];
                    sha256_rounds16<32>(st, wv);
                } else { sha256_schedule16(wv);
#pragma unroll
                         for (int i = 0; i < 16; ++i) xb0[i * 128 + node] = wv[i]; }
            }
            __syncthreads();
            // phase D: rounds 48-63, feed-forward, padding block
            Digest d0;
            if (active && is_main) {
#pragma unroll
                for (int i = 0; i < 16; ++i) wv[i] = xb0[i * 128 + node];
                sha256_rounds16<48>(st, wv);
#pragma unroll
                for (int i = 0; i < 8; ++i) d0.w[i] = st[i] + SHA_IV[i];
                uint32_t pad[16] = {0x80000000u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 512u};
                sha256_compress(d0.w, pad);
            }
            __syncthreads();                                       // every read of lvl and xch done
            if (active && is_main) { lds_store(&lvl[2 * node], d0); store_digest(nodes, out_base + node, d0); }
            __syncthreads();
            continue;
        }
        if (HASH == 1 && w > kFieldRowMaxNodes && w <= kFieldQuadMaxNodes) {
            // Level of 33 .. 64 nodes per workgroup of the field hash: one hash per QUAD of lanes (fieldhash_inner_quad_f64: a
            // 4-block of the state per lane, ~1 950 instructions, ~6 us), 64 hashes per pass of the workgroup.  Lane q reads children
            // words 4q .. 4q+3 as one 16-byte LDS access; lanes 0 and 1 of a quad hold the digest.
            const uint32_t q = tid & 3u, grp = tid >> 2;
            uint32_t res[2][4];
#pragma unroll
            for (uint32_t b = 0; b < 2; ++b) {
                if (b * 64 >= w) break;                                // workgroup-uniform
                const uint32_t node = b * 64 + grp;
                const uint4 v = lvl[4 * (node < w ? node : 0u) + q];
                const uint32_t in[4] = {v.x, v.y, v.z, v.w};
                fieldhash_inner_quad_f64(in, q, res[b], g_fh_consts64);
            }
            __syncthreads();                                       // every read of the level done
#pragma unroll
            for (uint32_t b = 0; b < 2; ++b) {
                if (b * 64 >= w) break;
                const uint32_t node = b * 64 + grp;
                if (node < w && q < 2) {
                    const uint4 o = make_uint4(res[b][0], res[b][1], res[b][2], res[b][3]);
                    lvl[2 * node + q] = o;
                    reinterpret_cast<uint4*>(nodes + (out_base + node) * 8)[q] = o;
                }
            }
            __syncthreads();
            continue;
        }
        if (HASH == 1 && w <= kFieldRowMaxNodes) {
            // Narrow level of the field hash: one hash per ROW of 16 lanes in double precision (fieldhash_inner_row16_f64: the 16
            // state elements on 16 lanes, quad broadcasts and row rotations, ~1 170 instructions), 16 hashes per pass of the
            // workgroup, 3.9 us per pass (the 32-bit row form of rounds 3-4: 5.4); the one-lane hash takes ~11 us for up to 256
            // nodes, so the row form pays up to 32 nodes per workgroup (profiles/r05_ab_field_row.txt: 16 / 32 / 64 swept).
            // Every lane runs the permutation (DPP needs whole rows).
            const uint32_t g = tid & 15u, grp = tid >> 4;
            uint32_t res[4];
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                if (b * 16 >= w) break;                                // workgroup-uniform
                const uint32_t node = b * 16 + grp;
                const uint32_t* words = reinterpret_cast<const uint32_t*>(&lvl[4 * (node < w ? node : 0u)]);   // children 2 node, 2 node + 1
                res[b] = fieldhash_inner_row16_f64(words[g], g, g_fh_consts64);
            }
            __syncthreads();                                       // every read of the level done
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                if (b * 16 >= w) break;
                const uint32_t node = b * 16 + grp;
                if (node < w && g < 8) {
                    reinterpret_cast<uint32_t*>(&lvl[2 * node])[g] = res[b];
                    nodes[(out_base + node) * 8 + g] = res[b];
                }
            }
            __syncthreads();
            continue;
        }
        // in place: node u is written over child slot u after every thread of the level has read
        Digest d0, d1;
        const bool a0 = tid < w, a1 = tid + kWgThreads < w;       // w <= 512: at most two nodes per thread
        if (a0) d0 = Hasher<HASH>::inner(lds_digest(&lvl[4 * tid]), lds_digest(&lvl[4 * tid + 2]));
        if (a1) d1 = Hasher<HASH>::inner(lds_digest(&lvl[4 * (tid + kWgThreads)]), lds_digest(&lvl[4 * (tid + kWgThreads) + 2]));
        __syncthreads();
        if (a0) { lds_store(&lvl[2 * tid], d0); store_digest(nodes, out_base + tid, d0); }
        if (a1) { lds_store(&lvl[2 * (tid + kWgThreads)], d1); store_digest(nodes, out_base + tid + kWgThreads, d1); }
        __syncthreads();
    }
    WG_STAMP((phase ? 24u : 2u) + lv);                           // end of the last level of this phase
    if (phase == 1 || j2 == 0) break;
    // Continuation.  Every workgroup has written its node of depth depth_in - j (the barrier that ends a level has drained
    // the stores of all its waves); an agent-scope release / acquire on a counter tells the one that arrives last, and that
    // one alone reads the gridDim.x nodes back (the acquire has invalidated its CU's vector L1) and carries on.
    __shared__ uint32_t go_on;
    if (gridDim.x > 1) {
        if (kWgRelaxedPublish) {
            // this workgroup's node (lvl[0..1] after the last level) goes out write-through; nothing else has to cross dies
            publish_node(nodes, (((size_t)1 << (depth_in - j)) - 1) + blockIdx.x, lvl, tid);
        } else {
            // every wave's digest stores have reached L2 before thread 0 releases them to the other compute dies: a wait on this
            // wave's own stores, then the barrier; the agent-scope release itself (an L2 write-back) is paid ONCE per workgroup,
            // by the atomic below (256 workgroups x 4 waves of agent-scope fences cost the SHA-256 path 4 us per tree)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (tid == 0) {
        uint32_t last = 1;
        if (gridDim.x > 1) {
            last = (kWgRelaxedPublish ? __hip_atomic_fetch_add(mail.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : __hip_atomic_fetch_add(mail.counter + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)) == gridDim.x - 1;
            if (last) __hip_atomic_store(mail.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        go_on = last;
    }
    __syncthreads();
    if (!go_on) return;                                            // workgroup-uniform
    // a launch that also posts layer values (once per proof): the values the OTHER workgroups stored (behind their __threadfence
    // above) are read by this one with ordinary loads at the end -- take them in with an acquire, as the counter did in rounds 1-5
    if (kWgRelaxedPublish && LEAF && mail.dump_src) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#ifdef ZK_WG_TRACE
    if (tid == 0) wg_rec = __hip_atomic_load(&g_wg_trace_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // workgroup 0 wrote it before it reached the counter
    WG_STAMP(15);
#endif
    d_in = depth_in - j; lv = j2; cnt = gridDim.x; first = 0;
    {
        const size_t base2 = ((size_t)1 << d_in) - 1;
        Digest dd[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u)
            if (tid + u * kWgThreads < cnt)
                dd[u] = (kWgRelaxedPublish && gridDim.x > 1) ? load_digest_agent(nodes, base2 + tid + u * kWgThreads) : load_digest(nodes, base2 + tid + u * kWgThreads);
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u)
            if (tid + u * kWgThreads < cnt) lds_store(&lvl[2 * (tid + u * kWgThreads)], dd[u]);
    }
    __syncthreads();
    WG_STAMP(16);
  }
    // The launch that reaches depth mail.top posts its 2^top digests (and, on request, the layer values) to the
    // host: small PCIe writes are slow, so the workgroup that finishes last copies everything with wide stores.
    if (mail.mailbox && depth_in - j - j2 == mail.top && off == 0) {
        __shared__ uint32_t is_last;
        // one node per workgroup crosses the dies write-through (publish_node); a launch that also posts layer values keeps the
        // release / acquire form, which covers them as well (once per proof: the layer that feeds the host's FRI tail)
        const bool relaxed = kWgRelaxedPublish && gridDim.x > 1 && j2 == 0 && mail.dump_src == nullptr;   // uniform over the launch
        if (relaxed) publish_node(nodes, (((size_t)1 << mail.top) - 1) + blockIdx.x, lvl, tid);
        if (tid == 0) {
            uint32_t last = 1;
            if (gridDim.x > 1 && j2 == 0) {                        // with a continuation only the last workgroup gets here
                // release: this workgroup's digest (and values) are out; acquire: so are everybody else's
                last = (relaxed ? __hip_atomic_fetch_add(mail.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : __hip_atomic_fetch_add(mail.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)) == gridDim.x - 1;
                if (last) __hip_atomic_store(mail.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            is_last = last;
        }
        __syncthreads();
        if (is_last) {
#ifdef ZK_WG_TRACE
            if (tid == 0 && wg_rec == 0xffffffffu) wg_rec = __hip_atomic_load(&g_wg_trace_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            WG_STAMP(40);
#endif
            const uint4* sq = reinterpret_cast<const uint4*>(nodes + (((size_t)1 << mail.top) - 1) * 8);
            uint4* dq = reinterpret_cast<uint4*>(mail.mailbox + kMailDigests);
            if (relaxed) {                                         // the peers' nodes: agent-scope loads (they are not in this die's L2)
                for (uint32_t i = tid; i < (1u << mail.top); i += kWgThreads) {
                    const Digest d = load_digest_agent(nodes, (((size_t)1 << mail.top) - 1) + i);
                    dq[2 * i] = make_uint4(d.w[0], d.w[1], d.w[2], d.w[3]);
                    dq[2 * i + 1] = make_uint4(d.w[4], d.w[5], d.w[6], d.w[7]);
                }
            } else {
                for (uint32_t i = tid; i < (2u << mail.top); i += kWgThreads) dq[i] = sq[i];
            }
            if (mail.dump_src) {
                const uint4* sv = reinterpret_cast<const uint4*>(mail.dump_src);
                uint4* dv = reinterpret_cast<uint4*>(mail.mailbox + mail.vals_off);
                for (uint32_t i = tid; i < (1u << mail.dump_log) / 4; i += kWgThreads) dv[i] = sv[i];
            }
            __threadfence_system();
            __syncthreads();
            WG_STAMP(41);
            if (tid == 0) __hip_atomic_store(&mail.mailbox[0], mail.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            WG_STAMP(42);
        }
    }
}

#ifdef ZK_WG_TRACE
// Appends the records collected so far to $ZK_WG_TRACE_FILE (one line per latency launch) and clears them; called when a context
// is destroyed.  meta: depth_in, j, j2, leaf, hash, hand-over depth + 1 (0: no mailbox), workgroups; then the stamps in 10 ns ticks
// relative to the launch's first stamp (0 = not reached).
void dump_wg_trace() {
    const char* path = getenv("ZK_WG_TRACE_FILE");
    unsigned int n = 0;
    if (!path || hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wg_trace_n), sizeof n) != hipSuccess || !n) return;
    if (n > kWgTraceLaunches) n = kWgTraceLaunches;
    std::vector<unsigned long long> rec((size_t)n * kWgTraceSlots);
    if (hipMemcpyFromSymbol(rec.data(), HIP_SYMBOL(g_wg_trace), rec.size() * 8) != hipSuccess) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    for (unsigned int r = 0; r < n; ++r) {
        const unsigned long long* x = &rec[(size_t)r * kWgTraceSlots];
        const unsigned long long m = x[0];
        fprintf(f, "wg depth_in=%llu j=%llu j2=%llu leaf=%llu hash=%llu top=%lld wgs=%llu abs0=%llu :", m & 255, (m >> 8) & 255, (m >> 16) & 255, (m >> 24) & 1, (m >> 25) & 1,
                (long long)((m >> 26) & 63) - 1, m >> 32, x[1]);
        for (uint32_t i = 1; i < kWgTraceSlots; ++i) fprintf(f, " %lld", x[i] ? (long long)(x[i] - x[1]) : -1LL);
        fprintf(f, "\n");
    }
    fclose(f);
    const unsigned int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_n), &zero, sizeof zero);
}
#endif

static double merkle_bytes(bool leaf, uint32_t depth, uint32_t k) {
    // inputs read once (4 B values or 32 B digests), every produced digest written once
    double in = (double)((size_t)1 << depth);
    double produced = (leaf ? in : 0.0) + in * (1.0 - 1.0 / (double)((size_t)1 << k));
    return (leaf ? 4.0 : 32.0) * in + 32.0 * produced;
}
static double merkle_ops(bool leaf, uint32_t depth, uint32_t k, int hash) {
    double in = (double)((size_t)1 << depth);
    double lo = hash ? kFieldLeafOps : kShaLeafOps, io = hash ? kFieldInnerOps : kShaInnerOps;
    return (leaf ? in * lo : 0.0) + in * (1.0 - 1.0 / (double)((size_t)1 << k)) * io;
}

static hipError_t ensure_fieldhash_consts() {
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || done[dev]) return hipSuccess;
    FieldHashConsts c;
    fieldhash_make_consts(c);
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_fh_consts), &c, sizeof c);
    if (e != hipSuccess) return e;
    static FieldHashConsts64 c64;
    fieldhash_make_consts64(c, c64);
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_fh_consts64), &c64, sizeof c64);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

// Throughput phase: subtree launches (k <= 4 levels each) while the level has more than 2^18 nodes,
// i.e. while there are more than ~4 waves per SIMD to keep busy.  Latency phase: workgroup launches
// of up to 10 levels each.
// Build-time constants (ZK_BUILD_DEFS="-DZK_MERKLE_MAX_K=3" to A/B them; profiles/r03_ab_max_k.txt, r03_config2_switches.txt):
// levels per subtree launch, and the level size at which the throughput phase hands over to the latency phase (measured
// flat optimum 2^16 .. 2^18 nodes).  The hand-over depth can also be moved at run time with zk_dev_set_merkle_latency_log
// (include/zkstark_amd.h: tests use it to drive small trees through the chunked build).
#ifndef ZK_MERKLE_MAX_K
#define ZK_MERKLE_MAX_K 4
#endif
#ifndef ZK_MERKLE_LATENCY_LOG
#define ZK_MERKLE_LATENCY_LOG 17
#endif
#ifndef ZK_MERKLE_CHUNK_K
#define ZK_MERKLE_CHUNK_K 3
#endif
static_assert(ZK_MERKLE_MAX_K >= 1 && ZK_MERKLE_MAX_K <= (int)kMerkleMaxK && ZK_MERKLE_CHUNK_K >= 1 && ZK_MERKLE_CHUNK_K <= (int)kMerkleMaxK, "levels per launch");
static_assert(ZK_MERKLE_LATENCY_LOG >= 12 && ZK_MERKLE_LATENCY_LOG <= 24, "latency switch");
static uint32_t g_merkle_latency_log = ZK_MERKLE_LATENCY_LOG;
static constexpr uint32_t merkle_max_k() { return ZK_MERKLE_MAX_K; }
static uint32_t merkle_latency_log() { return __atomic_load_n(&g_merkle_latency_log, __ATOMIC_RELAXED); }
bool set_merkle_latency_log(uint32_t v) {
    if (v == 0) v = ZK_MERKLE_LATENCY_LOG;
    if (v < 12 || v > 24) return false;
    __atomic_store_n(&g_merkle_latency_log, v, __ATOMIC_RELAXED);
    return true;
}

// ---- how the latency phase of a tree is cut into (phase 1, continuation) -------------------------------------------
// Microseconds one workgroup, alone on its compute unit, needs for a level of w nodes (every form costs one hash LATENCY per
// pass: measured per form): SHA-256 one lane per hash 4.6 (256 per pass), main / helper lanes 4.9 (128),
// four lanes per hash 3.1 (64); field hash one lane per hash in double precision ~11 (256 per pass; tools/fh64_probe.hip: 10.7 us
// on a lone wave), a row of 16 lanes in double precision 3.9 (16 per pass; the 32-bit row form of rounds 3-4: 5.4).
static double wg_level_us(uint32_t w, int hash) {
    if (w == 0) return 0.0;
    if (hash) return w <= kFieldRowMaxNodes ? (double)((w + 15) / 16) * 3.7 : w <= kFieldQuadMaxNodes ? (double)((w + 63) / 64) * 5.3 : (double)((w + 255) / 256) * 10.8;   // round 6: as traced (tools/wg_trace.py)
    return w <= 64 ? 3.1 : w <= 128 ? 4.9 : (double)((w + 255) / 256) * 4.6;
}
static double wg_phase_us(bool leaf, uint32_t cnt_log, uint32_t levels, int hash, uint32_t blocks) {
    const uint32_t cnt = 1u << cnt_log;
    double us = !leaf ? 1.0                                                        // the first load, or the leaf hashes:
                : hash ? (cnt <= kFieldRowLeafMax ? (double)((cnt + 15) / 16) * 3.7 : (double)((cnt + 255) / 256) * 10.8)
                       : (double)((cnt + 255) / 256) * 2.6;
    for (uint32_t t = 1; t <= levels; ++t) us += wg_level_us(cnt >> t, hash);
    return blocks > 256 ? us * (double)blocks / 256.0 : us;                        // more workgroups than compute units take turns
}
// Build-time A/B switch (ZK_BUILD_DEFS="-DZK_MERKLE_CONTINUATION=0": one launch per <= 10 levels, as rounds 1-4)
#ifndef ZK_MERKLE_CONTINUATION
#define ZK_MERKLE_CONTINUATION 1
#endif
static constexpr bool g_merkle_continuation = ZK_MERKLE_CONTINUATION != 0;
constexpr double kContinueUs = 6.0;    // the continuation's release -> count -> acquire -> reload
constexpr double kLaunchUs = 10.0;     // a further launch on the commit -> challenge -> launch path

// Builds the levels of a heap over 2^log_m leaves that lie above the aligned leaf range
// [chunk << log_sub, (chunk + 1) << log_sub), from the leaves (leaf_mode) or from the nodes already
// present at depth `log_m - log_sub + span` ... up to the chunk's own root at depth log_m - log_sub.
// A whole tree is chunk 0 with log_sub = log_m.
template <class SRC>
static hipError_t merkle_build_t(SRC src, double src_bytes, uint32_t log_m, uint32_t* nodes, hipStream_t s, Profiler* prof,
                                 const MailArgs& mail_in, int hash, uint32_t log_sub = 0xffffffffu, size_t chunk = 0,
                                 bool leaf_mode = true, uint32_t tp_floor = 0) {
    if (hash) {
        hipError_t e = ensure_fieldhash_consts();
        if (e != hipSuccess) return e;
    }
    if (log_sub == 0xffffffffu) log_sub = log_m;
    const uint32_t stop = log_m - log_sub;            // depth of the chunk root
    uint32_t depth = log_m;
    bool leaf = leaf_mode;
    const PlainSrc none{nullptr};
    // the first launch reads its leaves through SRC: replace the plain 4 B/leaf read by the source's bytes
    auto first_bytes = [&](double b) { return leaf ? b - 4.0 * (double)((size_t)1 << log_sub) + src_bytes : b; };
    const uint32_t kMerkleLatencyLog = merkle_latency_log();
    auto off_at = [&](uint32_t d) { return (size_t)chunk << (d - stop); };   // chunk's first node at depth d
    // a chunk build (tp_floor != 0) runs the subtree kernels down to that ABSOLUTE depth and stops there;
    // launch_merkle_finish builds everything above it once, for the whole tree
    const bool throughput_only = tp_floor != 0;
    const uint32_t floor_depth = throughput_only ? tp_floor : stop + kMerkleLatencyLog;
    while (depth > floor_depth) {
        uint32_t k = depth - floor_depth;
        if (k > merkle_max_k()) k = merkle_max_k();
        size_t lanes = (size_t)1 << (depth - stop - k);         // >= 2^17: a multiple of the block size
        uint32_t blocks = (uint32_t)(lanes / kMerkleThreads);
        size_t sh = (size_t)(kMerkleThreads / 64) * (k + 1) * 128 * sizeof(uint4);
        ScopedKernelTimer tm(prof, leaf ? K_MERKLE_LEAF : K_MERKLE_INNER, first_bytes(merkle_bytes(leaf, depth - stop, k)), s, merkle_ops(leaf, depth - stop, k, hash));
        if (hash) {
            if (leaf) hipLaunchKernelGGL((merkle_subtree_kernel<SRC, true, 1>), dim3(blocks), dim3(kMerkleThreads), sh, s, src, nodes, depth, k, off_at(depth));
            else hipLaunchKernelGGL((merkle_subtree_kernel<PlainSrc, false, 1>), dim3(blocks), dim3(kMerkleThreads), sh, s, none, nodes, depth, k, off_at(depth));
        } else {
            if (leaf) hipLaunchKernelGGL((merkle_subtree_kernel<SRC, true, 0>), dim3(blocks), dim3(kMerkleThreads), sh, s, src, nodes, depth, k, off_at(depth));
            else hipLaunchKernelGGL((merkle_subtree_kernel<PlainSrc, false, 0>), dim3(blocks), dim3(kMerkleThreads), sh, s, none, nodes, depth, k, off_at(depth));
        }
        leaf = false;
        depth -= k;
    }
    if (throughput_only) return hipGetLastError();
    // top > 0 (whole trees only): the build ends at depth `top`, whose 2^top digests go to the mailbox and the
    // levels above are hashed by the host (host_sha.hpp; batches: those digests are the per-proof roots)
    MailArgs mail = mail_in;
    if (stop != 0 || mail.top >= log_m) mail.top = 0;
    const uint32_t end = stop == 0 ? mail.top : stop;
    auto launch_wg = [&](uint32_t j, uint32_t j2) {
        const uint32_t span = depth - stop;
        const uint32_t blocks = 1u << (span - j);
        const uint32_t lds_log = j2 && span - j > j ? span - j : j;       // the continuation holds all `blocks` nodes
        const size_t sh = ((size_t)2 << lds_log) * sizeof(uint4) + 2 * 16 * 128 * sizeof(uint32_t);
        ScopedKernelTimer tm(prof, K_MERKLE_TOP, first_bytes(merkle_bytes(leaf, span, j)) + (j2 ? merkle_bytes(false, span - j, j2) : 0.0), s,
                             merkle_ops(leaf, span, j, hash) + (j2 ? merkle_ops(false, span - j, j2, hash) : 0.0));
        if (hash) {
            if (leaf) hipLaunchKernelGGL((merkle_wg_kernel<SRC, true, 1>), dim3(blocks), dim3(kWgThreads), sh, s, src, nodes, depth, j, mail, off_at(depth), j2, lds_log);
            else hipLaunchKernelGGL((merkle_wg_kernel<PlainSrc, false, 1>), dim3(blocks), dim3(kWgThreads), sh, s, none, nodes, depth, j, mail, off_at(depth), j2, lds_log);
        } else {
            if (leaf) hipLaunchKernelGGL((merkle_wg_kernel<SRC, true, 0>), dim3(blocks), dim3(kWgThreads), sh, s, src, nodes, depth, j, mail, off_at(depth), j2, lds_log);
            else hipLaunchKernelGGL((merkle_wg_kernel<PlainSrc, false, 0>), dim3(blocks), dim3(kWgThreads), sh, s, none, nodes, depth, j, mail, off_at(depth), j2, lds_log);
        }
        leaf = false;
        depth -= j + j2;
    };
    // A whole tree with a counter at hand finishes in ONE launch: every workgroup reduces 2^j inputs to a node and the one
    // that finishes last carries on with those nodes (merkle_wg_kernel: continuation).  (j, j2) is the split the level
    // cost model below likes best; a lone workgroup is a bad way to hash a small tree (2^10 leaves of the field hash:
    // 142 us in one workgroup, ~50 us as 32 workgroups and a continuation).
    const bool may_continue = stop == 0 && mail.counter != nullptr && g_merkle_continuation;
    do {
        const uint32_t span = depth - stop;
        const uint32_t levels = depth - end;
        // without a continuation: the remaining levels split evenly over launches of <= kWgMaxLog levels; this launch takes jn
        const uint32_t launches = levels ? (levels + kWgMaxLog - 1) / kWgMaxLog : 1;
        const uint32_t jn = (levels + launches - 1) / launches;
        uint32_t best_j = jn, best_j2 = 0;
        if (may_continue && levels <= 2 * kWgMaxLog) {
            // cost of the plain way to depth `end` (a further launch costs the host an enqueue on the commit -> challenge -> launch
            // path; only the two-launch case is compared, deeper trees take the plain way first)
            double best = wg_phase_us(leaf, jn, jn, hash, 1u << (span - jn));
            if (launches == 2 && span - jn <= kWgMaxLog) best += kLaunchUs + wg_phase_us(false, span - jn, levels - jn, hash, 1);
            else if (launches > 1) best = 0.0;                                  // not modelled: keep the plain way
            for (uint32_t j = 1; j < levels && j <= kWgMaxLog; ++j) {
                const uint32_t j2 = levels - j;
                if (j2 > kWgMaxLog || span - j > kWgMaxLog) continue;          // the continuation keeps 2^(span - j) nodes in LDS
                const double us = wg_phase_us(leaf, j, j, hash, 1u << (span - j)) + kContinueUs + wg_phase_us(false, span - j, j2, hash, 1);
                if (us < best) { best = us; best_j = j; best_j2 = j2; }
            }
        }
        launch_wg(best_j, best_j2);
    } while (depth > end);
    return hipGetLastError();
}

hipError_t launch_merkle_build(const uint32_t* vals, uint32_t log_m, uint32_t* nodes, hipStream_t s, Profiler* prof,
                               const MailArgs& mail, int hash) {
    return merkle_build_t(PlainSrc{vals}, 4.0 * (double)((size_t)1 << log_m), log_m, nodes, s, prof, mail, hash);
}
// commitment of a block whose leaves are still in all-to-all order (no interleave pass, no block buffer)
hipError_t launch_merkle_build_interleaved(const uint32_t* recv, uint32_t log_parts, uint32_t log_cnt, uint32_t* nodes, hipStream_t s,
                                           Profiler* prof, int hash, const MailArgs& mail) {
    uint32_t log_m = log_parts + log_cnt;
    return merkle_build_t(InterleaveSrc{recv, log_parts, log_cnt}, 4.0 * (double)((size_t)1 << log_m), log_m, nodes, s, prof, mail, hash);
}
// One chunk (1 / 2^log_chunks of the leaves, still in all-to-all order in its own receive buffer) of a
// tree over 2^log_m leaves: levels up to the chunk root.  launch_merkle_finish joins the chunk roots.
// Depth at which chunk builds hand over to launch_merkle_finish.  A chunk build is ONE leaf launch (the leaves of the
// chunk and k <= 4 levels, straight from its receive buffer: that is the part worth overlapping with the exchange of the
// next chunk); everything above -- further throughput launches and the latency phase -- runs once over the whole tree,
// at full width.  (Round 2 let every chunk run its own inner launches down to the latency switch: four quarter-width
// launches per level group, each a quarter-filled chip.)  Trees too small for that hand over at the chunk roots.
// A chunk build takes k = 3 levels, not 4: a chunk of 2^22 leaves is then 2 048 workgroups of 32 KiB of LDS (1.6 rounds
// of five per CU) instead of 1 024 of 40 KiB that start and end together, so LDS frees up all through the launch and the
// workgroups of the exchange kernel that runs beside it (rcclGenericKernel: 64 x 256 threads, 19.5 KiB of LDS each,
// measured) find room on the CUs.  The hashing rate does not depend on k (profiles/r03_ab_max_k.txt).
static constexpr uint32_t merkle_chunk_k() { return ZK_MERKLE_CHUNK_K; }
static uint32_t chunk_handover_depth(uint32_t log_m, uint32_t log_chunks) {
    const uint32_t lat = merkle_latency_log();
    if (!(log_m > lat && lat >= log_chunks + 8)) return log_chunks;
    uint32_t k = log_m - lat < merkle_max_k() ? log_m - lat : merkle_max_k();
    if (k > merkle_chunk_k()) k = merkle_chunk_k();
    return log_m - k;
}
uint32_t merkle_finish_start_depth(uint32_t log_m, uint32_t log_chunks) { return chunk_handover_depth(log_m, log_chunks); }
hipError_t launch_merkle_build_chunk(const uint32_t* recv, uint32_t log_parts, uint32_t log_cnt, uint32_t* nodes, uint32_t log_m,
                                     uint32_t chunk, hipStream_t s, Profiler* prof, int hash) {
    uint32_t log_sub = log_parts + log_cnt;
    const uint32_t h = chunk_handover_depth(log_m, log_m - log_sub);
    return merkle_build_t(InterleaveSrc{recv, log_parts, log_cnt}, 4.0 * (double)((size_t)1 << log_sub), log_m, nodes, s, prof, MailArgs{}, hash,
                          log_sub, chunk, true, h != log_m - log_sub ? h : 0u);
}
// After every chunk of a 2^log_m-leaf tree (2^log_chunks chunks) has been built down to the hand-over depth:
// the remaining throughput levels and the latency phase of the whole tree, once.
hipError_t launch_merkle_finish(uint32_t* nodes, uint32_t log_m, uint32_t log_chunks, hipStream_t s, Profiler* prof, int hash,
                                const MailArgs& mail) {
    const uint32_t start = chunk_handover_depth(log_m, log_chunks);   // depth the chunk builds stopped at
    if (start == 0) return hipSuccess;
    // the nodes at depth `start` exist: inner mode over a "tree" of 2^start inputs shares the top of the heap
    return merkle_build_t(PlainSrc{nullptr}, 0.0, start, nodes, s, prof, mail, hash, start, 0, false);
}
// fold + commit of the folded layer (a.out receives it): one pass
hipError_t launch_fold_merkle(const FoldArgs& a, uint32_t* nodes, hipStream_t s, Profiler* prof, const MailArgs& mail, int hash) {
    uint32_t log_out = a.log_m - 1;
    return merkle_build_t(FoldSrc{a}, 12.0 * (double)((size_t)1 << log_out), log_out, nodes, s, prof, mail, hash);
}
// composition + commit of cp layer 0 (a.cp receives it): one pass
hipError_t launch_compose_merkle(const ComposeArgs& a, uint32_t* nodes, hipStream_t s, Profiler* prof, const MailArgs& mail, int hash) {
    return merkle_build_t(ComposeSrc{a}, 8.0 * (double)((size_t)1 << a.logN), a.logN, nodes, s, prof, mail, hash);
}

// composition over one rank's block of a sharded proof + the subtree over it: reads the block of f (4 bytes per leaf; the
// taps hit the same lines), the range table (4), writes nothing but the tree
hipError_t launch_compose_block_merkle(const ComposeBlockArgs& b, uint32_t* nodes, hipStream_t s, Profiler* prof, const MailArgs& mail, int hash) {
    return merkle_build_t(ComposeBlockSrc{b}, 8.0 * (double)((size_t)1 << b.log_m), b.log_m, nodes, s, prof, mail, hash);
}

hipError_t launch_compose_merkle_batch(const ComposeBatchArgs& a, uint32_t log_batch, uint32_t* nodes, hipStream_t s, Profiler* prof,
                                       const MailArgs& mail, int hash) {
    uint32_t log_m = a.a.logN + log_batch;
    return merkle_build_t(ComposeBatchSrc{a.a, a.chal}, 8.0 * (double)((size_t)1 << log_m), log_m, nodes, s, prof, mail, hash);
}
hipError_t launch_fold_merkle_batch(const FoldBatchArgs& a, uint32_t log_batch, uint32_t* nodes, hipStream_t s, Profiler* prof,
                                    const MailArgs& mail, int hash) {
    uint32_t log_out = a.a.log_m - 1 + log_batch;
    return merkle_build_t(FoldBatchSrc{a.a, a.chal}, 12.0 * (double)((size_t)1 << log_out), log_out, nodes, s, prof, mail, hash);
}

// Copies host-built pieces (tree tops, small layers) from the mapped staging buffer into the device
// arrays so that the device state is complete after a proof (zk_merkle_path, zk_layer_read).
__global__ __launch_bounds__(1024) void scatter_kernel(const uint32_t* stage, const ScatterSeg* segs, uint32_t* trees, uint32_t* layers) {
    const ScatterSeg sg = segs[blockIdx.x];
    uint32_t* dst = (sg.kind ? layers : trees) + sg.dst;
    const uint32_t* src = stage + sg.src;
    if (((sg.src | sg.dst | sg.words) & 3) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (uint32_t i = threadIdx.x; i < sg.words / 4; i += blockDim.x) d4[i] = s4[i];
    } else {
        for (uint32_t i = threadIdx.x; i < sg.words; i += blockDim.x) dst[i] = src[i];
    }
}
hipError_t launch_scatter(const uint32_t* stage, const ScatterSeg* segs, uint32_t count, double words, uint32_t* trees, uint32_t* layers,
                          hipStream_t s, Profiler* prof) {
    if (!count) return hipSuccess;
    ScopedKernelTimer tm(prof, K_GATHER, 8.0 * words, s);
    hipLaunchKernelGGL(scatter_kernel, dim3(count), dim3(1024), 0, s, stage, segs, trees, layers);
    return hipGetLastError();
}

// ===========================================================================
// Roofline probe of the Merkle kernels (measurement only; bench.py: roofline.valu.chain_*)
// ===========================================================================
// Every lane runs a chain of `hashes` inner hashes d <- H(d, d ^ seed) (merkle.rs:42-45 shape), no memory traffic:
// the rate at which the COMPILED hash issues, which is what the subtree kernels can reach at best at the same
// residency.  Each wave records its lifetime in shader clocks and in the 100 MHz constant clock (the clock held).
template <int HASH>
__global__ __launch_bounds__(256) void hash_chain_probe_kernel(uint32_t* out, uint32_t seed, uint32_t hashes, unsigned long long* rec) {
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; ++i) d.w[i] = (seed * (uint32_t)(i + 1) + threadIdx.x + blockIdx.x * 977u) % P;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (uint32_t it = 0; it < hashes; ++it) {
        Digest r = d;
        r.w[0] ^= seed;
        if (HASH) r.w[0] &= 0x7fffffffu;          // stays a canonical residue
        d = Hasher<HASH>::inner(d, r);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= d.w[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (rec && (threadIdx.x & 63) == 0) {
        rec[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = c1 - c0;
        rec[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}
hipError_t launch_hash_chain_probe(int hash, uint32_t blocks, uint32_t* out, uint32_t seed, uint32_t hashes, unsigned long long* rec, hipStream_t s) {
    if (hash) {
        hipError_t e = ensure_fieldhash_consts();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(hash_chain_probe_kernel<1>, dim3(blocks), dim3(256), 0, s, out, seed, hashes, rec);
    } else {
        hipLaunchKernelGGL(hash_chain_probe_kernel<0>, dim3(blocks), dim3(256), 0, s, out, seed, hashes, rec);
    }
    return hipGetLastError();
}

// The three forms of the field hash on the device against each other (test hook: zk_probe_fieldhash_forms).  Thread t hashes a
// pseudo-random pair of digests and a leaf value, every 16th thread one of the edge patterns (words 0, P - 1, all P - 1, all 0,
// raw words >= P): double precision (what every tree is built with) and 32-bit Montgomery (the host's form) must agree word for
// word and be canonical; the 16-lane row form of the narrow levels is compared on the first pair of each row.
__global__ __launch_bounds__(256) void fieldhash_forms_kernel(uint32_t seed, uint32_t* bad, uint32_t* first_bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = (t + 1u) * 2654435761u + seed;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    Digest l, r;
#pragma unroll
    for (int i = 0; i < 8; ++i) { l.w[i] = rnd() % P; r.w[i] = rnd() % P; }
    if ((t & 15u) == 1) { l.w[t & 7u] = 0; r.w[(t >> 4) & 7u] = P - 1; }
    if ((t & 15u) == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { l.w[i] = P - 1; r.w[i] = P - 1; }
    }
    if ((t & 15u) == 3) { l.w[0] = 0xFFFFFFFFu; r.w[7] = P; r.w[3] = P + 5; }                  // raw words >= P (field.rs:20-24)
    if ((t & 15u) == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { l.w[i] = 0; r.w[i] = 0; }
    }
    uint32_t v = rnd();
    if ((t & 7u) == 0) v = (t & 8u) ? 0xFFFFFFFFu : P - 1;
    if (t == 5) v = 0;
    const Digest a = fieldhash_inner(l, r, g_fh_consts), b = fieldhash_inner64(l, r, g_fh_consts64);
    const Digest c = fieldhash_leaf(v, g_fh_consts), d = fieldhash_leaf64(v, g_fh_consts64);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) ok = ok && a.w[i] == b.w[i] && c.w[i] == d.w[i] && b.w[i] < P && d.w[i] < P;
    // row form: the 16 lanes of a row hash the pair of the row's first thread
    const uint32_t g = threadIdx.x & 15u;
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t lw = __shfl(l.w[i], (int)(threadIdx.x & ~15u) & 63, 64), rw = __shfl(r.w[i], (int)(threadIdx.x & ~15u) & 63, 64);
        if (g == (uint32_t)i) word = lw;
        if (g == (uint32_t)i + 8u) word = rw;
    }
    {   // quad form: the 4 lanes of a quad hash the pair of the quad's first thread
        const uint32_t q4 = threadIdx.x & 3u;
        const int src = (int)(threadIdx.x & ~3u) & 63;
        uint32_t in4[4], out4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t wsel = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t lw = __shfl(l.w[i], src, 64), rw = __shfl(r.w[i], src, 64);
                if ((int)(4 * q4) + j == i) wsel = lw;
                if ((int)(4 * q4) + j == i + 8) wsel = rw;
            }
            in4[j] = wsel;
        }
        fieldhash_inner_quad_f64(in4, q4, out4, g_fh_consts64);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t want = __shfl(b.w[i], src, 64);
            if (q4 < 2 && (int)(4 * q4) <= i && i < (int)(4 * q4) + 4 && out4[i - 4 * (int)q4 < 0 ? 0 : (i - 4 * (int)q4) & 3] != want) ok = false;
        }
    }
    const uint32_t row = fieldhash_inner_row16(word, g, g_fh_consts), row64 = fieldhash_inner_row16_f64(word, g, g_fh_consts64);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t want = __shfl(b.w[i], (int)(threadIdx.x & ~15u) & 63, 64);
        if (g == (uint32_t)i && (row != want || row64 != want)) ok = false;
    }
    {   // Directed inputs of the double-precision S-box (fieldhash_f64.hpp: the s_0 path of the partial rounds squares up to 2^41.9,
        // where the quotient of the first product needs all 53 bits): integers of magnitude 2^37.8 ... 2^42 and multiples of P next to
        // them, both signs, against x^5 (mod P) in plain integer arithmetic.
        const double mags[6] = {236118324143.0 /* 2^37.78 */, 3829546563953.0 /* 2^41.8 */, 4104397301943.0 /* 2^41.9 */, 4398046511103.0 /* 2^42 - 1 */,
                                (double)P * 1365.0, (double)P * 1024.0};
        const uint32_t k = t % 6u;
        double xd = mags[k] + (double)(int)((rnd() & 0xFFFFu)) - 32768.0 + (k >= 4 ? (double)(rnd() % 3u) - 1.0 : 0.0);
        if (t & 8u) xd = -xd;
        const long long xi = (long long)xd;                           // exact: |x| < 2^43
        const uint64_t res = (uint64_t)(((xi % (long long)P) + (long long)P) % (long long)P);
        const uint64_t x2 = res * res % P, x4 = x2 * x2 % P, want5 = x4 * res % P;
        if (fh64_canonical(fh64_sbox(xd)) != (uint32_t)want5) ok = false;
        // and a whole partial round from a state whose s_0 sits at that magnitude: every output against integer arithmetic
        double st[kFhT];
        uint64_t ref[kFhT];
#pragma unroll
        for (int i = 0; i < kFhT; ++i) { st[i] = i ? (double)(l.w[i & 7] >> 1) - (double)(r.w[i & 7] >> 2) : xd; }
#pragma unroll
        for (int i = 0; i < kFhT; ++i) { const long long v = (long long)st[i]; ref[i] = (uint64_t)(((v % (long long)P) + (long long)P) % (long long)P); }
        const double rc = g_fh_consts64.rc_part[t % kFhRP];
        fh64_partial_round<true>(st, rc);
        {
            const uint64_t y = (ref[0] + (uint64_t)rc) % P, y2 = y * y % P, y4 = y2 * y2 % P;
            ref[0] = y4 * y % P;
            uint64_t sum = 0;
#pragma unroll
            for (int i = 0; i < kFhT; ++i) sum = (sum + ref[i]) % P;
            uint64_t outw[kFhT];
            outw[0] = (sum + 2 * (P - ref[0])) % P;                   // d_0 = -2
#pragma unroll
            for (int i = 1; i < kFhT; ++i) outw[i] = (((uint64_t)1 << (i - 1)) % P * ref[i] + sum) % P;
#pragma unroll
            for (int i = 0; i < kFhT; ++i)
                if (fh64_canonical(st[i]) != (uint32_t)outw[i]) ok = false;
        }
    }
    if (!ok) { atomicAdd(bad, 1u); atomicMin(first_bad, t); }
}
hipError_t launch_fieldhash_forms(uint32_t blocks, uint32_t seed, uint32_t* d_res, hipStream_t s) {
    hipError_t e = ensure_fieldhash_consts();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fieldhash_forms_kernel, dim3(blocks), dim3(256), 0, s, seed, d_res, d_res + 1);
    return hipGetLastError();
}

// ===========================================================================
// Trace generation for batches
// ===========================================================================
// prover.rs:32-39 is a serial recurrence, so one trace cannot be parallelised; many independent
// traces can.  One lane per trace: out[t*count + i] = a_i of trace t (a0[t], a1[t] seeds).
__global__ __launch_bounds__(64) void trace_fibsq_batch_kernel(const uint32_t* a0, const uint32_t* a1, uint32_t batch,
                                                               uint32_t count, uint32_t* out, uint32_t stride) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch) return;
    uint32_t* row = out + (size_t)t * stride;
    // Montgomery domain: squares of Montgomery values stay Montgomery
    uint32_t x = mont_mul(a0[t], R2_MONT), y = mont_mul(a1[t], R2_MONT);
    if (count > 0) row[0] = mont_mul(x, 1u);
    if (count > 1) row[1] = mont_mul(y, 1u);
    for (uint32_t i = 2; i < count; ++i) {
        uint32_t z = add(mont_mul(x, x), mont_mul(y, y));
        row[i] = mont_mul(z, 1u);
        x = y; y = z;
    }
}

hipError_t launch_trace_fibsq_batch(const uint32_t* a0, const uint32_t* a1, uint32_t batch, uint32_t count, uint32_t* out, hipStream_t s,
                                    uint32_t stride) {
    if (!batch) return hipSuccess;
    hipLaunchKernelGGL(trace_fibsq_batch_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, a0, a1, batch, count, out, stride ? stride : count);
    return hipGetLastError();
}

// ===========================================================================
// Interleave (cyclic <-> block layout change around the multi-GPU exchange)
// ===========================================================================
// in: 2^log_parts pieces of 2^log_cnt words; out[u * parts + q] = in[q * cnt + u]
__global__ void interleave_kernel(const uint32_t* in, uint32_t* out, uint32_t log_parts, uint32_t log_cnt) {
    size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= ((size_t)1 << (log_parts + log_cnt))) return;
    size_t q = o & (((size_t)1 << log_parts) - 1), u = o >> log_parts;
    out[o] = in[(q << log_cnt) + u];
}

__global__ __launch_bounds__(256) void halo_pack_kernel(const uint32_t* loc, uint32_t* out, uint32_t log_per, uint32_t lg, uint32_t h) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (h << lg)) return;
    const uint32_t q = i / h, u = i - q * h;
    const size_t mask = ((size_t)1 << (log_per + lg)) - 1;
    out[i] = loc[((((size_t)q + 1) << log_per) + u) & mask];
}
hipError_t launch_halo_pack(const uint32_t* loc, uint32_t* out, uint32_t log_per, uint32_t lg, uint32_t h, hipStream_t s) {
    const uint32_t total = h << lg;
    hipLaunchKernelGGL(halo_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, loc, out, log_per, lg, h);
    return hipGetLastError();
}
hipError_t launch_interleave(const uint32_t* in, uint32_t* out, uint32_t log_parts, uint32_t log_cnt, hipStream_t s) {
    size_t total = (size_t)1 << (log_parts + log_cnt);
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    hipLaunchKernelGGL(interleave_kernel, dim3(blocks), dim3(256), 0, s, in, out, log_parts, log_cnt);
    return hipGetLastError();
}

// ---- known-pattern exchange (self-test of the sharded prover's transport, shard.hip) ----------------------------
__global__ void pattern_fill_kernel(uint32_t* dst, uint32_t rank, uint32_t log_per, uint32_t log_parts) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << (log_per + log_parts))) return;
    dst[i] = shard_pattern(rank, (uint32_t)(i >> log_per), (uint32_t)(i & (((size_t)1 << log_per) - 1)));
}
__global__ void pattern_check_kernel(const uint32_t* src, uint32_t rank, uint32_t log_per, uint32_t log_parts, uint32_t log_chunks, uint32_t* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << (log_per + log_parts))) return;
    const uint32_t log_cc = log_per - log_chunks;
    const uint32_t u = (uint32_t)(i & (((size_t)1 << log_cc) - 1));
    const uint32_t q = (uint32_t)(i >> log_cc) & ((1u << log_parts) - 1u), c = (uint32_t)(i >> (log_cc + log_parts));
    if (src[i] != shard_pattern(q, rank, (c << log_cc) | u)) {
        atomicAdd(&out[0], 1u);
        atomicMin(&out[1], (uint32_t)i);
    }
}
hipError_t launch_pattern_fill(uint32_t* dst, uint32_t rank, uint32_t log_per, uint32_t log_parts, hipStream_t s) {
    const size_t total = (size_t)1 << (log_per + log_parts);
    hipLaunchKernelGGL(pattern_fill_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, dst, rank, log_per, log_parts);
    return hipGetLastError();
}
hipError_t launch_pattern_check(const uint32_t* src, uint32_t rank, uint32_t log_per, uint32_t log_parts, uint32_t log_chunks, uint32_t* out, hipStream_t s) {
    const size_t total = (size_t)1 << (log_per + log_parts);
    hipLaunchKernelGGL(pattern_check_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, src, rank, log_per, log_parts, log_chunks, out);
    return hipGetLastError();
}

// ===========================================================================
// Decommit gather
// ===========================================================================
__global__ void gather_kernel(const uint32_t* src, const uint64_t* offsets, uint32_t count, uint32_t words, uint32_t* out) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * words) return;
    uint32_t item = idx / words, w = idx % words;
    out[idx] = src[offsets[item] + w];
}

hipError_t launch_gather(const uint32_t* src, const uint64_t* offsets, uint32_t count, uint32_t words,
                         uint32_t* out, hipStream_t s, Profiler* prof) {
    if (!count) return hipSuccess;
    ScopedKernelTimer tm(prof, K_GATHER, 8.0 * (double)count * words, s);
    uint32_t total = count * words, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, s, src, offsets, count, words, out);
    return hipGetLastError();
}


// One launch for a whole decommitment: work list and results in host-mapped memory, flag behind the results.
constexpr int kFetchThreads = 1024;
__global__ __launch_bounds__(kFetchThreads) void fetch_kernel(const uint32_t* layers, const uint32_t* trees, const uint64_t* items, uint32_t nv,
                                                              uint32_t ndg, uint32_t* out, uint32_t* mailbox, uint32_t seq, uint32_t* counter) {
    const uint32_t stride = gridDim.x * kFetchThreads;
    for (uint32_t idx = blockIdx.x * kFetchThreads + threadIdx.x; idx < 2 * ndg + nv; idx += stride) {
        if (idx < 2 * ndg) {                                      // half a node per thread: 16-byte accesses on both sides
            const uint4* src = reinterpret_cast<const uint4*>(trees + items[nv + (idx >> 1)]);
            reinterpret_cast<uint4*>(out)[idx] = src[idx & 1u];
        } else {
            const uint32_t i = idx - 2 * ndg;
            out[8 * (size_t)ndg + i] = layers[items[i]];
        }
    }
    __threadfence_system();                                       // this thread's results are in host memory ...
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = true;
        if (gridDim.x > 1) {                                      // ... and so are everybody else's
            last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
            if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) __hip_atomic_store(&mailbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_fetch(const uint32_t* layers, const uint32_t* trees, const uint64_t* items, uint32_t nv, uint32_t ndg, uint32_t* out,
                        uint32_t* mailbox, uint32_t seq, uint32_t* counter, hipStream_t s, Profiler* prof) {
    ScopedKernelTimer tm(prof, K_GATHER, 8.0 * ((double)nv + 8.0 * (double)ndg), s);
    const uint32_t work = 2 * ndg + nv;
    uint32_t blocks = (work + kFetchThreads - 1) / kFetchThreads;
    if (blocks == 0) blocks = 1;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(fetch_kernel, dim3(blocks), dim3(kFetchThreads), 0, s, layers, trees, items, nv, ndg, out, mailbox, seq, counter);
    return hipGetLastError();
}

}  // namespace zk
