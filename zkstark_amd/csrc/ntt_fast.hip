// ntt_fast.hip -- register-radix NTT pass (R = 32, 64, 128, 256; radix 512 was built and measured slower in round 5:
// profiles/r05_ab_ntt_radix512.txt, removed in round 6).
//
// Same pass semantics as ntt_pass_kernel (kernels.hip): the array is [A][R][S], a workgroup owns
// C columns x R rows.  The tile is R*C = 4096 words, 2048 for transforms of up to 2^20 words
// (NttPassArgs.tile_log; kernels.hpp: ntt_tile_log says why: many light workgroups per compute unit).
// The R-point transform is a four-step inside the tile, R = Ra*Rb with Ra <= 16, Rb <= 32:
//
//   step 1  thread (tb, c): loads rows ta*Rb + tb, ta < Ra, straight into registers (lanes run
//           along c: whole 128 B row segments), applies the inter-pass twiddle
//           as a running product, runs the Ra-point DFT in registers with compile-time
//           twiddles (17 multiplies for 16 points, 5 for 8), multiplies by w_R^(tb*ka) and
//           writes row ka*Rb + tb of the LDS tile;
//   step 2  thread (ka, c): reads rows ka*Rb + tb, Rb-point DFT in registers, stores rows
//           ka + Ra*kb straight to HBM (again whole row segments).
//
// One LDS round trip and one barrier per pass instead of log2(R); ~3.8 Montgomery products per
// element per pass instead of ~9.  Passes whose inner stride S is smaller than C (the innermost
// pass and the first LDE pass) stage their contiguous tile through LDS so that HBM accesses stay
// full lines.  No bit reversal anywhere: register naming absorbs it.
//
// Addressing (round 4).  A pass is VALU-bound in this field (P > 2^31: every add / sub carries its
// correction), so address arithmetic on the vector unit is paid for in butterflies: the ISA of round 3
// spent 20 % of its VALU instructions on 64-bit element addresses (v_lshlrev_b64, v_lshl_add_u64, v_or).
// Here every global access is a raw buffer access: (the workgroup's base in a buffer resource) + (uniform row
// offset in an SGPR, computed on the scalar unit) + (32-bit byte offset of the lane in one VGPR, computed once
// per thread): `buffer_load_dword v, v_off, s[rsrc], s_row offen`, no vector instruction per access.  (Plain
// pointers do not get there: hipcc re-associates (base + row) + lane into a 64-bit vector add per access.)
#include "kernels.hpp"

#include "field.hpp"

namespace zk {

namespace {

constexpr uint32_t c_mulmod(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); }
constexpr uint32_t c_powmod(uint32_t a, uint64_t e) {
    uint32_t r = 1, b = a;
    while (e) {
        if (e & 1) r = c_mulmod(r, b);
        b = c_mulmod(b, b);
        e >>= 1;
    }
    return r;
}
constexpr uint32_t c_to_mont(uint32_t a) { return (uint32_t)((((uint64_t)a) << 32) % P); }
constexpr uint32_t kW32 = c_powmod(GEN_W, (uint64_t)(P - 1) >> 5);          // primitive 32nd root of unity
constexpr uint32_t kW32Inv = c_powmod(kW32, 31);

// w32^e (forward) or w32^-e (inverse) in Montgomery form, e < 16
template <bool INV>
struct Root32 {
    static constexpr uint32_t w(int e) { return c_to_mont(c_powmod(INV ? kW32Inv : kW32, (uint64_t)e)); }
};

constexpr int c_brev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// In-register DFT of 2^LOG points, radix-2 decimation in frequency: natural order in,
// x[i] = X[bitrev(i)] out.  All indices and twiddles are compile-time constants.
template <bool INV, int LOG>
__device__ __forceinline__ void dft_regs(uint32_t (&x)[1 << LOG]) {
    constexpr int R = 1 << LOG;
#pragma unroll
    for (int ll = LOG - 1; ll >= 0; --ll) {
#pragma unroll
        for (int b = 0; b < R / 2; ++b) {
            const int len = 1 << ll;
            const int j = b & (len - 1);
            const int i = ((b >> ll) << (ll + 1)) | j;
            const int e = (j << (LOG - 1 - ll)) * (32 / R);      // exponent of w32
            uint32_t u = x[i], v = x[i + len];
            x[i] = add(u, v);
            uint32_t d = sub(u, v);
            x[i + len] = (e == 0) ? d : mont_mul(d, Root32<INV>::w(e));
        }
    }
}

__device__ __forceinline__ uint32_t pow_lookup(const PowTable& t, uint32_t e) {
    return mont_mul(t.hi[e >> t.lo_bits], t.lo[e & ((1u << t.lo_bits) - 1u)]);
}
// the two table reads of pow_lookup, issued where the exponent is known; the product is taken where it is needed
struct PowRaw { uint32_t hi, lo; };
__device__ __forceinline__ PowRaw pow_fetch(const PowTable& t, uint32_t e) { return PowRaw{t.hi[e >> t.lo_bits], t.lo[e & ((1u << t.lo_bits) - 1u)]}; }
__device__ __forceinline__ uint32_t pow_of(const PowRaw& r) { return mont_mul(r.hi, r.lo); }

// Raw buffer over [base, base + 4 GiB): word at base + lane_off + row_off (bytes; lane_off in a VGPR, row_off uniform).
// An arrays of this library has at most 2^30 words, so every offset fits 32 bits; out-of-range reads return 0.
using Rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ Rsrc make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
// A uniform value pinned to a scalar register where it is used.  Without this the register allocator parks long-lived
// row offsets in VGPRs when SGPRs run short, and a buffer instruction whose scalar offset sits in a VGPR becomes a
// waterfall loop (v_readfirstlane + compare + branch per access); readfirstlane of a value that already is in an SGPR folds away.
__device__ __forceinline__ uint32_t in_sgpr(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t ld_b(Rsrc r, uint32_t lane_off, uint32_t row_off) { return __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, 0); }
__device__ __forceinline__ void st_b(Rsrc r, uint32_t lane_off, uint32_t row_off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, lane_off, row_off, 0); }

constexpr int kThreads = 256;

// MODE: NTT_DIF (inverse passes: DFT with w^-1, post-twiddle), NTT_DIT (forward passes: pre-twiddle),
//       NTT_DIT_LDE (forward, S = B, source = n coefficients, each feeding B columns).
// STAGED: the tile is one contiguous block of HBM (S < C): go through LDS for full-line accesses.
// PREP (NTT_DIT_LDE only): the source is the raw DIF output U of the inverse transform and the
//   interpolant's coefficient preparation (coef_prepare_kernel: virtual last trace point, coset shift,
//   1/n) happens inside this pass -- one launch and one sweep over the coefficients less.
//   PREP = 1: every value is prepared as a column loads it (B-fold redundant arithmetic; kept for B = 1, where
//             nothing is redundant and the staging area below would be as large as the tile);
//   PREP = 2: the tile's C / B coefficient blocks (contiguous in the source) are loaded ONCE, prepared and kept in
//             LDS; the B columns of a block read them from there.  One preparation per coefficient at every size,
//             so the separate sweep of round 3 (12 us at n = 2^21) is gone from every LDE.
template <uint32_t MODE, int LA, int LB, bool STAGED, int TILE_LOG, int PREP>
__global__ __launch_bounds__(kThreads) void ntt_pass_fast_kernel(NttPassArgs p) {
    constexpr bool INV = (MODE == NTT_DIF);
    constexpr bool LDE = (MODE == NTT_DIT_LDE);
    constexpr int LOGR = LA + LB, R = 1 << LOGR, RA = 1 << LA, RB = 1 << LB;
    constexpr int LOGC = TILE_LOG - LOGR, C = 1 << LOGC;
    constexpr uint32_t kTileLog = TILE_LOG;
    constexpr int PITCH = STAGED || LDE ? C + 1 : C;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    p.src += (size_t)blockIdx.y * p.src_stride;      // batch of independent transforms
    p.dst += (size_t)blockIdx.y * p.dst_stride;
    uint32_t* tile = smem;                    // R * PITCH
    uint32_t* twl = smem + R * PITCH;         // w_R^e, e < R
    uint32_t* coef = twl + R;                 // PREP = 2: (C >> logB) blocks of R prepared coefficients, pitch R + 1
    const uint32_t tid = threadIdx.x;
    const uint32_t logS = p.logS;
    const uint32_t col0 = blockIdx.x << LOGC;
    const uint32_t smask = (1u << logS) - 1u;
    const uint32_t tw_shift = p.L - LOGR - logS;

    // A small transform runs one or two workgroups per compute unit, and a pass is then a chain of memory round trips
    // (twiddle tables, data, twiddles of the second step): every read whose address is known here is issued here, so
    // that they all wait together.  (RB * C == kThreads for every tile in use: a thread has ONE step-1 item.)
    constexpr bool ONE = RB * C == kThreads;
    constexpr bool PRELOAD = ONE && !STAGED && !LDE;
    constexpr int IT2 = (RA * C + kThreads - 1) / kThreads;   // RA*C < kThreads (small tile, R = 256): half the threads idle in step 2
    uint32_t xpre[RA];
    PowRaw pre_cur{}, pre_step{}, post_cur[IT2], post_step[IT2];
    if (ONE) {
        const uint32_t c = tid & (C - 1), tb = tid >> LOGC;
        if (PRELOAD) {
            const uint32_t off = (tb << (logS + 2)) + (c << 2);
            const Rsrc b = make_rsrc(p.src + (((size_t)(col0 >> logS) << (LOGR + logS)) + (col0 & smask)));
#pragma unroll
            for (int ta = 0; ta < RA; ++ta) xpre[ta] = ld_b(b, off, in_sgpr((uint32_t)(ta * RB) << (logS + 2)));
        }
        if (MODE != NTT_DIF && logS) {
            const uint32_t s1 = (col0 + c) & smask;
            pre_cur = pow_fetch(p.tw, (tb * s1) << tw_shift);
            pre_step = pow_fetch(p.tw, ((uint32_t)RB * s1) << tw_shift);
        }
    }
    if (MODE == NTT_DIF && logS) {
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const uint32_t q = tid + it * kThreads;
            const uint32_t s2 = (col0 + (q & (C - 1))) & smask, ka = (q >> LOGC) & (RA - 1);
            post_cur[it] = pow_fetch(p.tw, (ka * s2) << tw_shift);
            post_step[it] = pow_fetch(p.tw, ((uint32_t)RA * s2) << tw_shift);
        }
    }
    for (uint32_t e = tid; e < (uint32_t)R; e += kThreads) twl[e] = pow_lookup(p.tw, e << (p.L - LOGR));

    // Non-staged tiles (S >= C) lie inside one block `a` of the [A][R][S] view: element (row t, column c) of the tile is
    // word  (a << (LOGR + logS)) + (t << logS) + s0 + c.  The first and third terms are uniform (the workgroup's base),
    // the row term is uniform per unrolled ta / kb (scalar add), the lane term (tb or ka, c) is one VGPR.
    const uint32_t row_sh = logS + 2;                                        // log2 of the bytes between two rows
    const size_t wg_word = ((size_t)(col0 >> logS) << (LOGR + logS)) + (col0 & smask);
    const Rsrc srcb = make_rsrc(p.src + wg_word), dstb = make_rsrc(p.dst + wg_word);
    // contiguous tiles (STAGED, and the destination of the first LDE pass)
    const Rsrc tsrc = make_rsrc(p.src + ((size_t)blockIdx.x << kTileLog)), tdst = make_rsrc(p.dst + ((size_t)blockIdx.x << kTileLog));

    if (STAGED && !LDE) {
        // the tile is R*C consecutive words: copy in memory order, lanes along the fastest index
#pragma unroll 4
        for (uint32_t l = tid; l < (uint32_t)(R * C); l += kThreads) {
            uint32_t s = l & smask, t = (l >> logS) & (R - 1), c = ((l >> (LOGR + logS)) << logS) | s;
            tile[t * PITCH + c] = ld_b(tsrc, l << 2, 0);
        }
    }
    if (LDE && PREP == 2) {
        // the coefficient blocks col0 >> logB .. of this tile are (C >> logB) * R consecutive words of the source.
        // Storage position (a, t) holds the coefficient of true index k = rev(a) | t << (log_n - LOGR): the slow storage
        // digits of the digit-reversed DIF output are the LOW digits of k.
        //   out = (U - U[n-1] g^(k+1)) * shift^k / n        (coef_prepare_kernel, kernels.hip)
        const uint32_t logB = logS, log_n = p.prep_log_n, tsh = log_n - LOGR, n1 = (1u << log_n) - 1u;
        const uint32_t a0 = col0 >> logB;
        const Rsrc csrc = make_rsrc(p.src + ((size_t)a0 << LOGR));
        const uint32_t c_top = p.src[n1];                          // U[n-1]
        for (uint32_t l = tid; l < ((uint32_t)C >> logB) << LOGR; l += kThreads) {
            const uint32_t ab = l >> LOGR, t = l & (R - 1), a = a0 + ab;
            uint32_t ka = 0, rem = tsh, sh = 0;
#pragma unroll                                                     // constant indices: the argument struct stays in SGPRs (no scratch copy)
            for (uint32_t d = 0; d + 1 < (uint32_t)kMaxDigits; ++d) {
                const uint32_t bits = d + 1 < p.prep_nd ? p.prep_bits[d] : 0u;
                rem -= bits;
                ka |= ((a >> rem) & ((1u << bits) - 1u)) << sh;
                sh += bits;
            }
            const uint32_t k = ka | (t << tsh);
            const uint32_t v = ld_b(csrc, l << 2, 0);
            const uint32_t gk = pow_lookup(p.tw, ((k + 1u) & n1) << p.prep_log_b);
            const uint32_t wk = mont_mul(pow_lookup(p.prep_wtab, k), p.prep_ninv_mont);
            coef[ab * (R + 1) + t] = mont_mul(sub(v, mont_mul(c_top, gk)), wk);
        }
    }
    __syncthreads();   // twl (and the staged tile / the prepared coefficients) visible

    // ---- step 1: Ra-point DFTs over ta (row stride Rb) ---------------------------------
#pragma unroll 1
    for (uint32_t q = tid; q < (uint32_t)(RB * C); q += kThreads) {
        const uint32_t c = q & (C - 1), tb = q >> LOGC;
        const uint32_t s = (col0 + c) & smask;
        uint32_t x[RA];
        if (LDE && PREP == 2) {
            const uint32_t* cb = coef + (c >> logS) * (R + 1) + tb;     // B columns share a block: broadcast reads, blocks in different banks
#pragma unroll
            for (int ta = 0; ta < RA; ++ta) x[ta] = cb[ta * RB];
        } else if (LDE) {
            const uint32_t a = (col0 + c) >> logS;                 // coefficient block; B columns share it
            const Rsrc cb = make_rsrc(p.src);
            const uint32_t off = (a << (LOGR + 2)) | (tb << 2);     // n <= 2^30 words: the byte offset fits 32 bits
#pragma unroll
            for (int ta = 0; ta < RA; ++ta) x[ta] = ld_b(cb, off, (uint32_t)(ta * RB * 4));
            if (PREP == 1) {
                const uint32_t log_n = p.prep_log_n, tsh = log_n - LOGR;
                uint32_t ka = 0, rem = tsh, sh = 0;
#pragma unroll
                for (uint32_t d = 0; d + 1 < (uint32_t)kMaxDigits; ++d) {
                    const uint32_t bits = d + 1 < p.prep_nd ? p.prep_bits[d] : 0u;
                    rem -= bits;
                    ka |= ((a >> rem) & ((1u << bits) - 1u)) << sh;
                    sh += bits;
                }
                const uint32_t n1 = (1u << log_n) - 1u;
                const uint32_t k0 = ka | (tb << tsh), kstep = (uint32_t)RB << tsh;
                const uint32_t c_top = p.src[n1];                  // U[n-1]
                // x <- (x - U[n-1] g^(k+1)) * shift^k / n, running products over ta (k advances by kstep)
                uint32_t gcur = pow_lookup(p.tw, ((k0 + 1u) & n1) << p.prep_log_b);
                uint32_t wcur = mont_mul(pow_lookup(p.prep_wtab, k0), p.prep_ninv_mont);
                const uint32_t gstep = pow_lookup(p.tw, (kstep & n1) << p.prep_log_b), wstep = pow_lookup(p.prep_wtab, kstep & n1);
#pragma unroll
                for (int ta = 0; ta < RA; ++ta) {
                    x[ta] = mont_mul(sub(x[ta], mont_mul(c_top, gcur)), wcur);
                    if (ta + 1 < RA) { gcur = mont_mul(gcur, gstep); wcur = mont_mul(wcur, wstep); }
                }
            }
        } else if (STAGED) {
#pragma unroll
            for (int ta = 0; ta < RA; ++ta) x[ta] = tile[(ta * RB + tb) * PITCH + c];
        } else if (PRELOAD) {
#pragma unroll
            for (int ta = 0; ta < RA; ++ta) x[ta] = xpre[ta];
        } else {
            const uint32_t off = (tb << row_sh) + (c << 2);
#pragma unroll
            for (int ta = 0; ta < RA; ++ta) x[ta] = ld_b(srcb, off, in_sgpr((uint32_t)(ta * RB) << row_sh));
        }
        if (MODE != NTT_DIF && logS) {
            // pre-twiddle w_{RS}^(t*s), t = ta*Rb + tb: running product over ta
            uint32_t cur = ONE ? pow_of(pre_cur) : pow_lookup(p.tw, (tb * s) << tw_shift);
            const uint32_t step = ONE ? pow_of(pre_step) : pow_lookup(p.tw, ((uint32_t)RB * s) << tw_shift);
#pragma unroll
            for (int ta = 0; ta < RA; ++ta) {
                x[ta] = mont_mul(x[ta], cur);
                if (ta + 1 < RA) cur = mont_mul(cur, step);
            }
        }
        // (staged tile: this DFT reads and rewrites only rows = tb (mod Rb) of column c, which no other
        //  DFT of step 1 touches, so no barrier is needed between its loads and its stores)
        dft_regs<INV, LA>(x);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int ka = c_brev(i, LA);
            uint32_t v = x[i];
            if (ka != 0) v = mont_mul(v, twl[tb * ka]);
            tile[(ka * RB + tb) * PITCH + c] = v;
        }
    }
    __syncthreads();

    // ---- step 2: Rb-point DFTs over tb (consecutive rows) ----------------------------------
#ifndef ZK_NTT_LDE_DIRECT
#define ZK_NTT_LDE_DIRECT 1
#endif
    // The first LDE pass writes rows of S = B words: B >= 8 makes them whole 32-byte sectors, which go straight to HBM
    // like the rows of a non-staged pass instead of through the LDS tile (two barriers and two LDS sweeps less: 36.4 ->
    // 33.3 us at N = 2^24, profiles/r04_ab_ntt_lde.txt; ZK_BUILD_DEFS="-DZK_NTT_LDE_DIRECT=0" is the other side of that A/B).
    const bool lde_direct = ZK_NTT_LDE_DIRECT && LDE && logS >= 3;
    constexpr bool TO_LDS = STAGED || LDE;                     // contiguous destination tile: stores go through LDS
    uint32_t keep[IT2][RB];   // staged stores wait until every thread has read the tile
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
        const uint32_t q = tid + it * kThreads;
        if (RA * C < kThreads && q >= (uint32_t)(RA * C)) continue;
        const uint32_t c = q & (C - 1), ka = q >> LOGC;
        uint32_t (&y)[RB] = keep[it];
#pragma unroll
        for (int tb = 0; tb < RB; ++tb) y[tb] = tile[(ka * RB + tb) * PITCH + c];
        dft_regs<INV, LB>(y);
        if (MODE == NTT_DIF) {
            if (logS) {
                // post-twiddle w_{RS}^(k*s), k = ka + Ra*kb: running product over kb
                uint32_t cur = pow_of(post_cur[it]);
                const uint32_t step = pow_of(post_step[it]);
#pragma unroll
                for (int kb = 0; kb < RB; ++kb) {
                    const int i = c_brev(kb, LB);
                    y[i] = mont_mul(y[i], cur);
                    if (kb + 1 < RB) cur = mont_mul(cur, step);
                }
            } else if (p.scale_mont) {
#pragma unroll
                for (int i = 0; i < RB; ++i) y[i] = mont_mul(y[i], p.scale_mont);
            }
        }
        if (!TO_LDS) {
            const uint32_t off = (ka << row_sh) + (c << 2);
#pragma unroll
            for (int i = 0; i < RB; ++i) st_b(dstb, off, in_sgpr((uint32_t)(RA * c_brev(i, LB)) << row_sh), y[i]);
        } else if (ZK_NTT_LDE_DIRECT && LDE && lde_direct) {
            // column c of the tile = block a = c >> logS, position s = c & smask of the [A][R][S] view (col0 is a multiple of C >= S)
            const uint32_t off = ((c >> logS) << (LOGR + logS + 2)) + (ka << row_sh) + ((c & smask) << 2);
#pragma unroll
            for (int i = 0; i < RB; ++i) st_b(tdst, off, in_sgpr((uint32_t)(RA * c_brev(i, LB)) << row_sh), y[i]);
        }
    }
    if (TO_LDS && !lde_direct) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const uint32_t q = tid + it * kThreads;
            if (RA * C < kThreads && q >= (uint32_t)(RA * C)) continue;
            const uint32_t c = q & (C - 1), ka = q >> LOGC;
#pragma unroll
            for (int i = 0; i < RB; ++i) tile[(ka + RA * c_brev(i, LB)) * PITCH + c] = keep[it][i];
        }
        __syncthreads();
#pragma unroll 4
        for (uint32_t l = tid; l < (uint32_t)(R * C); l += kThreads) {
            uint32_t s = l & smask, t = (l >> logS) & (R - 1), c = ((l >> (LOGR + logS)) << logS) | s;
            st_b(tdst, l << 2, 0, tile[t * PITCH + c]);
        }
    }
}

template <uint32_t MODE, int LA, int LB, int TILE_LOG>
hipError_t launch3(const NttPassArgs& a, bool staged, uint32_t blocks, hipStream_t s) {
    constexpr int R = 1 << (LA + LB), C = 1 << (TILE_LOG - LA - LB);
    size_t shmem = ((size_t)R * (C + 1) + R) * sizeof(uint32_t);
    const dim3 grid(blocks, a.batch ? a.batch : 1);
    if (MODE == NTT_DIT_LDE && a.prep) {
        if (a.logS >= 1) {
            shmem += (size_t)(C >> a.logS) * (R + 1) * sizeof(uint32_t);
            hipLaunchKernelGGL((ntt_pass_fast_kernel<MODE, LA, LB, true, TILE_LOG, MODE == NTT_DIT_LDE ? 2 : 0>), grid, dim3(kThreads), shmem, s, a);
        } else {
            hipLaunchKernelGGL((ntt_pass_fast_kernel<MODE, LA, LB, true, TILE_LOG, MODE == NTT_DIT_LDE ? 1 : 0>), grid, dim3(kThreads), shmem, s, a);
        }
    }
    else if (staged) hipLaunchKernelGGL((ntt_pass_fast_kernel<MODE, LA, LB, true, TILE_LOG, 0>), grid, dim3(kThreads), shmem, s, a);
    else hipLaunchKernelGGL((ntt_pass_fast_kernel<MODE, LA, LB, false, TILE_LOG, 0>), grid, dim3(kThreads), shmem, s, a);
    return hipGetLastError();
}

template <uint32_t MODE, int LA, int LB>
hipError_t launch2(const NttPassArgs& a, bool staged, uint32_t blocks, hipStream_t s) {
    return a.tile_log == kSmallTileLog ? launch3<MODE, LA, LB, (int)kSmallTileLog>(a, staged, blocks, s)
                                       : launch3<MODE, LA, LB, (int)kMidTileLog>(a, staged, blocks, s);
}

template <uint32_t MODE>
hipError_t launch1(const NttPassArgs& a, bool staged, uint32_t blocks, hipStream_t s) {
    switch (a.logR) {
        case 5: return launch2<MODE, 3, 2>(a, staged, blocks, s);
        case 6: return launch2<MODE, 3, 3>(a, staged, blocks, s);
        case 7: return launch2<MODE, 4, 3>(a, staged, blocks, s);
        case 8: return launch2<MODE, 4, 4>(a, staged, blocks, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// True when launch_ntt_pass_fast will take this pass (the planner asks before it fuses the coefficient
// preparation into the first LDE pass, which only this kernel implements).
bool ntt_fast_ok(const NttPassArgs& a, NttMode mode) {
    if (a.logR < 5 || a.logR > 8) return false;
    if (a.tile_log != kSmallTileLog && a.tile_log != kMidTileLog) return false;
    const uint32_t logC = a.tile_log - a.logR;
    if (a.log_total < a.tile_log || a.logC != logC) return false;
    if (mode == NTT_DIT_LDE && !(a.logS < logC)) return false;
    return true;
}

// Returns true if this pass was launched on the fast path (full tiles only).
bool launch_ntt_pass_fast(const NttPassArgs& a, NttMode mode, hipStream_t s, hipError_t* err) {
    if (!ntt_fast_ok(a, mode)) return false;
    const bool staged = a.logS < a.tile_log - a.logR;
    const uint32_t blocks = 1u << (a.log_total - a.tile_log);
    switch (mode) {
        case NTT_DIF: *err = launch1<NTT_DIF>(a, staged, blocks, s); break;
        case NTT_DIT: *err = launch1<NTT_DIT>(a, staged, blocks, s); break;
        case NTT_DIT_LDE: *err = launch1<NTT_DIT_LDE>(a, true, blocks, s); break;
    }
    return true;
}

}  // namespace zk
