// fh64_probe.hip -- is the field hash faster in double precision?  (csrc/fieldhash_f64.hpp against csrc/fieldhash.hpp)
//   1. equality: 2^20 random inner hashes and leaf hashes (edge words 0, P-1, raw words >= P included) through both forms;
//   2. issue rates of the double-precision ops the new form is made of (v_fma_f64, v_add_f64, v_mul_f64, v_rndne_f64);
//   3. the two compiled hashes in a dependent chain, 1 .. 8 workgroups per CU, 10 launches back to back: ns per hash per SIMD
//      (one workgroup per CU = one wave per SIMD: the latency of one hash on a lone wave);
//   4. the two 16-lane row forms (narrow tree levels) in a dependent chain.
// Build: hipcc -O3 --offload-arch=gfx950 -I zkstark_amd/csrc -o tools/fh64_probe tools/fh64_probe.hip zkstark_amd/csrc/host_sha.cpp
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "fieldhash_f64.hpp"

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace zk;

__constant__ FieldHashConsts g_c32;
__constant__ FieldHashConsts64 g_c64;

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__global__ void check_kernel(uint32_t* bad, uint32_t* first) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = t * 2654435761u + 12345u;
    Digest l, r;
    for (int i = 0; i < 8; ++i) { l.w[i] = rnd(s) % P; r.w[i] = rnd(s) % P; }
    if ((t & 15u) == 1) { l.w[t & 7u] = 0; r.w[(t >> 3) & 7u] = P - 1; }
    if ((t & 15u) == 2) { for (int i = 0; i < 8; ++i) { l.w[i] = P - 1; r.w[i] = P - 1; } }
    if ((t & 15u) == 3) { l.w[0] = 0xFFFFFFFFu; r.w[7] = P; r.w[3] = P + 5; }            // raw words >= P
    if ((t & 15u) == 4) { for (int i = 0; i < 8; ++i) { l.w[i] = 0; r.w[i] = 0; } }
    const Digest a = fieldhash_inner(l, r, g_c32), b = fieldhash_inner64(l, r, g_c64);
    uint32_t v = rnd(s);
    if ((t & 7u) == 0) v = (t & 8u) ? 0xFFFFFFFFu : P - 1;
    if (t == 5) v = 0;
    const Digest c = fieldhash_leaf(v, g_c32), d = fieldhash_leaf64(v, g_c64);
    bool ok = true;
    for (int i = 0; i < 8; ++i) ok = ok && a.w[i] == b.w[i] && c.w[i] == d.w[i] && b.w[i] < P && d.w[i] < P;
    if (!ok) { atomicAdd(bad, 1u); atomicMin(first, t); }
}

template <int F64>
__global__ __launch_bounds__(256) void chain_kernel(uint32_t* out, uint32_t seed, int hashes) {
    Digest d;
#pragma unroll
    for (int i = 0; i < 8; ++i) d.w[i] = (seed * (i + 1) + threadIdx.x + blockIdx.x * 977u) % P;
#pragma unroll 1
    for (int it = 0; it < hashes; ++it) {
        Digest r = d;
        r.w[0] ^= 1u;
        d = F64 ? fieldhash_inner64(d, r, g_c64) : fieldhash_inner(d, r, g_c32);
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= d.w[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

// the 16-lane row forms in a dependent chain: lane g of a row holds word g of left || right; the digest comes back in lanes g < 8
template <int F64>
__global__ __launch_bounds__(256) void row_chain_kernel(uint32_t* out, uint32_t seed, int hashes) {
    const uint32_t g = threadIdx.x & 15u;
    uint32_t w = (seed * (g + 1) + (threadIdx.x >> 4) * 131u + blockIdx.x * 977u) % P;
#pragma unroll 1
    for (int it = 0; it < hashes; ++it) {
        const uint32_t d = F64 ? fieldhash_inner_row16_f64(w, g, g_c64) : fieldhash_inner_row16(w, g, g_c32);
        const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x120 + 8, 0xF, 0xF, false);   // lanes 8..15 take lanes 0..7
        w = g < 8 ? d : (up ^ 1u) % P;
    }
    out[blockIdx.x * 256 + threadIdx.x] = w;
}

constexpr int ITER = 128, UNROLL = 32, ACC = 8;
template <int OP>
__global__ __launch_bounds__(256) void op_kernel(double* out, double seed) {
    double a[ACC], b[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) { a[i] = seed * (i + 1) + threadIdx.x; b[i] = seed * 0.5 + (i + 3) + threadIdx.x * 7; }
    const double k = 1.0000001, m = 0.5;
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < ACC; ++i) {
                const double y = b[(i + u) & (ACC - 1)];
                if (OP == 0) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(k), "v"(m));
                if (OP == 1) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(y));
                if (OP == 2) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(k));
                if (OP == 3) asm volatile("v_rndne_f64 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            }
        }
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch(r + 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    FieldHashConsts c32;
    fieldhash_make_consts(c32);
    static FieldHashConsts64 c64;
    fieldhash_make_consts64(c32, c64);
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_c32), &c32, sizeof c32));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_c64), &c64, sizeof c64));
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
    uint32_t* d_flag;
    CHK(hipMalloc(&d_flag, 8));
    uint32_t init[2] = {0u, 0xFFFFFFFFu};
    CHK(hipMemcpy(d_flag, init, 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, d_flag, d_flag + 1);
    CHK(hipDeviceSynchronize());
    uint32_t res[2];
    CHK(hipMemcpy(res, d_flag, 8, hipMemcpyDeviceToHost));
    printf("equality: %u of %u (inner, leaf) pairs differ%s\n", res[0], 4096u * 256u, res[0] ? "  <-- MISMATCH" : " (bit-identical, digests canonical)");
    if (res[0]) printf("first differing thread %u\n", res[1]);
    hipFuncAttributes fa;
    CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(chain_kernel<0>)));
    printf("chain kernel, 32-bit Montgomery: %d VGPRs, %zu B scratch\n", fa.numRegs, (size_t)fa.localSizeBytes);
    CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(chain_kernel<1>)));
    printf("chain kernel, double precision : %d VGPRs, %zu B scratch\n", fa.numRegs, (size_t)fa.localSizeBytes);
    const char* names[4] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_rndne_f64"};
    for (int op = 0; op < 4; ++op)
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps, reps = 5;
            double ms = time_ms([&](int r) {
                if (op == 0) hipLaunchKernelGGL(op_kernel<0>, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<double*>(d_out), 1.5 + r);
                if (op == 1) hipLaunchKernelGGL(op_kernel<1>, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<double*>(d_out), 1.5 + r);
                if (op == 2) hipLaunchKernelGGL(op_kernel<2>, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<double*>(d_out), 1.5 + r);
                if (op == 3) hipLaunchKernelGGL(op_kernel<3>, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<double*>(d_out), 1.5 + r);
            }, reps);
            const double instr = (double)reps * wps * ITER * UNROLL * ACC;      // per SIMD (one wave of a block per SIMD)
            printf("%-12s %d waves/SIMD: %6.3f ns per instruction per SIMD (%.2f cycles at 2.4 GHz)\n", names[op], wps, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
        }
    // (a form of the double-precision hash with the 16 S-boxes of a full round advancing in step, for lone waves, was measured
    //  here in round 5: 10.52 against 10.61 us per hash at one wave per SIMD -- the compiler's schedule is not what limits it; removed)
    const char* forms[2] = {"32-bit Montgomery", "double precision"};
    for (int f64 = 0; f64 < 2; ++f64)
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps, reps = 10, hashes = 8;
            double ms = time_ms([&](int r) {
                if (f64) hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_out, 77u + r, hashes);
                else hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_out, 77u + r, hashes);
            }, reps);
            printf("inner hash chain, %-18s %d waves/SIMD launched: %8.2f ns per hash per SIMD%s\n", forms[f64], wps,
                   ms * 1e6 / ((double)reps * wps * hashes), wps == 1 ? "  (= the latency of one hash on a lone wave)" : "");
        }
    for (int f64 = 0; f64 < 2; ++f64)
        for (int wps : {1, 2}) {
            const int blocks = cus * wps, reps = 10, hashes = 16;
            double ms = time_ms([&](int r) {
                if (f64) hipLaunchKernelGGL(row_chain_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_out, 99u + r, hashes);
                else hipLaunchKernelGGL(row_chain_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_out, 99u + r, hashes);
            }, reps);
            printf("16-lane row form, %-18s %d waves/SIMD launched: %8.2f ns per dependent hash\n", f64 ? "double precision" : "32-bit Montgomery", wps,
                   ms * 1e6 / ((double)reps * hashes));
        }
    return 0;
}
