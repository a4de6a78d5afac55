// sha_ilp_probe.hip -- would hashing TWO (or four) independent nodes per lane, with their instruction streams
// interleaved round by round, let the two-cycle VALU ops of SHA-256 (v_bitop3_b32, v_add_u32: 38 % of the mix) pair up
// across waves?  A wave whose next instruction is always independent of its last never stalls on a result, so every
// wave on the SIMD always offers an instruction; tools/valu_mix_probe.hip reaches 3.3 cycles per instruction that way
// with a synthetic mix, the compiled single-hash stream 3.9-4.1 (tools/sha_latency_probe.hip).
// Build: hipcc -O3 --offload-arch=gfx950 -I zkstark_amd/csrc -o tools/sha_ilp_probe tools/sha_ilp_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "sha256.hpp"

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace zk;
constexpr int CH = 32;
constexpr double kValuPerInner = 2262.0;

// C interleaved compressions: the same operation is issued for every chain before the next operation starts
template <int C>
__device__ __forceinline__ void compress_multi(uint32_t (&st)[C][8], uint32_t (&w)[C][16]) {
    uint32_t a[C], b[C], c[C], d[C], e[C], f[C], g[C], h[C];
#pragma unroll
    for (int k = 0; k < C; ++k) { a[k] = st[k][0]; b[k] = st[k][1]; c[k] = st[k][2]; d[k] = st[k][3]; e[k] = st[k][4]; f[k] = st[k][5]; g[k] = st[k][6]; h[k] = st[k][7]; }
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        uint32_t wi[C];
        if (i < 16) {
#pragma unroll
            for (int k = 0; k < C; ++k) wi[k] = w[k][i];
        } else {
            uint32_t s0[C], s1[C];
#pragma unroll
            for (int k = 0; k < C; ++k) { uint32_t x = w[k][(i - 15) & 15]; s0[k] = sha_xor3(sha_rotr(x, 7), sha_rotr(x, 18), x >> 3); }
#pragma unroll
            for (int k = 0; k < C; ++k) { uint32_t x = w[k][(i - 2) & 15]; s1[k] = sha_xor3(sha_rotr(x, 17), sha_rotr(x, 19), x >> 10); }
#pragma unroll
            for (int k = 0; k < C; ++k) { wi[k] = (w[k][i & 15] + s0[k] + w[k][(i - 7) & 15]) + s1[k]; w[k][i & 15] = wi[k]; }
        }
        uint32_t S1[C], t1[C], S0[C], mj[C];
#pragma unroll
        for (int k = 0; k < C; ++k) S1[k] = sha_xor3(sha_rotr(e[k], 6), sha_rotr(e[k], 11), sha_rotr(e[k], 25));
#pragma unroll
        for (int k = 0; k < C; ++k) t1[k] = (h[k] + S1[k] + sha_ch(e[k], f[k], g[k])) + (SHA_K[i] + wi[k]);
#pragma unroll
        for (int k = 0; k < C; ++k) S0[k] = sha_xor3(sha_rotr(a[k], 2), sha_rotr(a[k], 13), sha_rotr(a[k], 22));
#pragma unroll
        for (int k = 0; k < C; ++k) mj[k] = sha_maj(a[k], b[k], c[k]);
#pragma unroll
        for (int k = 0; k < C; ++k) { h[k] = g[k]; g[k] = f[k]; f[k] = e[k]; e[k] = d[k] + t1[k]; d[k] = c[k]; c[k] = b[k]; b[k] = a[k]; a[k] = t1[k] + S0[k] + mj[k]; }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) { st[k][0] += a[k]; st[k][1] += b[k]; st[k][2] += c[k]; st[k][3] += d[k]; st[k][4] += e[k]; st[k][5] += f[k]; st[k][6] += g[k]; st[k][7] += h[k]; }
}

template <int C>
__device__ __forceinline__ void inner_multi(Digest (&d)[C], uint32_t seed) {
    uint32_t st[C][8], w[C][16];
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[k][i] = d[k].w[i]; w[k][8 + i] = d[k].w[i] ^ (i == 0 ? seed : 0u); st[k][i] = SHA_IV[i]; }
    compress_multi<C>(st, w);
    uint32_t pad[C][16];
#pragma unroll
    for (int k = 0; k < C; ++k) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pad[k][i] = 0;
        pad[k][0] = 0x80000000u; pad[k][15] = 512u;
    }
    compress_multi<C>(st, pad);
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) d[k].w[i] = st[k][i];
}

template <int C>
__global__ __launch_bounds__(256) void probe(uint32_t* out, uint32_t seed, unsigned long long* rec) {
    Digest d[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) d[c].w[i] = seed * (i + 1 + 8 * c) + threadIdx.x + blockIdx.x * 977;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < CH; ++it) inner_multi<C>(d, seed);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = 0;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) x ^= d[c].w[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) {
        rec[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = c1 - c0;
        rec[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}

template <int C>
int run(int cus, int wps, uint32_t* d_out, unsigned long long* d_rec) {
    const int blocks = cus * wps;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<C>, dim3(blocks), dim3(256), 0, 0, d_out, 3u, d_rec);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<C>, dim3(blocks), dim3(256), 0, 0, d_out, 5u, d_rec);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 8);
    CHK(hipMemcpy(h.data(), d_rec, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (size_t i = 0; i < h.size(); i += 2) ghz.push_back((double)h[i] / (double)h[i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double g = ghz[ghz.size() / 2];
    const double wall_cpi = ms * 1e-3 * g * 1e9 / ((double)wps * CH * C * kValuPerInner);
    printf("%d interleaved hashes per lane, %d wave(s) per SIMD: %.2f cycles per VALU instruction per SIMD by wall time (%.2f GHz, kernel %.1f us, %.2f ns per hash per SIMD)\n",
           C, wps, wall_cpi, g, ms * 1e3, ms * 1e6 / ((double)wps * CH * C));
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d_out; unsigned long long* d_rec;
    CHK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
    CHK(hipMalloc(&d_rec, (size_t)cus * 8 * 4 * 16));
    for (int wps : {1, 2, 4, 8}) if (run<1>(cus, wps, d_out, d_rec)) return 1;
    for (int wps : {1, 2, 4}) if (run<2>(cus, wps, d_out, d_rec)) return 1;
    for (int wps : {1, 2}) if (run<4>(cus, wps, d_out, d_rec)) return 1;
    return 0;
}
