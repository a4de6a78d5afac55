// sha_latency_probe.hip -- how long ONE dependent SHA-256 inner hash takes on a wave, which is what bounds the
// latency phase of a Merkle build (levels with fewer nodes than the chip has lanes: DESIGN.md section 4.3).
// Every lane runs a chain of CH inner hashes d <- H(d, d ^ c) (merkle.rs:42-45 shape: two compressions, the second
// on constant padding); waves record their s_memtime lifetime.  Residency 1 and 2 waves per SIMD; variant B hashes
// two independent chains per lane (does instruction-level parallelism inside one wave buy anything?).
// Build: hipcc -O3 --offload-arch=gfx950 -I zkstark_amd/csrc -o tools/sha_latency_probe tools/sha_latency_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "sha256.hpp"

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace zk;
constexpr int CH = 64;            // hashes per chain: a launch is 0.3 - 2.5 ms, the launch overhead disappears in it
constexpr double kValuPerInner = 2262.0;   // VALU instructions of one compiled sha256_inner (ISA count: 940 v_alignbit_b32, 597 v_bitop3_b32,
                                           // 365 v_add3_u32, 262 v_add_u32, 90 v_lshrrev_b32, 8 other)

template <int CHAINS>
__global__ __launch_bounds__(256) void probe(uint32_t* out, uint32_t seed, unsigned long long* rec) {
    Digest d[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) d[c].w[i] = seed * (i + 1 + 8 * c) + threadIdx.x + blockIdx.x * 977;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < CH; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            Digest r = d[c];
            r.w[0] ^= seed;
            d[c] = sha256_inner(d[c], r);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) x ^= d[c].w[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) {
        rec[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = c1 - c0;
        rec[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}

template <int CHAINS>
int run(int cus, int wps, uint32_t* d_out, unsigned long long* d_rec) {
    const int blocks = cus * wps;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<CHAINS>, dim3(blocks), dim3(256), 0, 0, d_out, 3u, d_rec);
    CHK(hipDeviceSynchronize());
    // Wall time over kReps launches BACK TO BACK.  Round 2 timed ONE launch: its 256 x wps workgroups do not spread evenly
    // over the 256 CUs, the kernel ends with the most loaded CU, and the figure carried that residency tail (4.0 cycles
    // per instruction where the steady state is 3.5: the review caught it against tools/valu_mix_probe2.hip, which
    // always timed ten launches).  Back to back, the next launch fills the CUs that finish early.
    constexpr int kReps = 10;
    CHK(hipEventRecord(e0));
    for (int r = 0; r < kReps; ++r) hipLaunchKernelGGL(probe<CHAINS>, dim3(blocks), dim3(256), 0, 0, d_out, 5u + r, d_rec);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= kReps;
    std::vector<unsigned long long> h((size_t)blocks * 8);
    CHK(hipMemcpy(h.data(), d_rec, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, ghz;
    for (size_t i = 0; i < h.size(); i += 2) { cyc.push_back((double)h[i] / (CH * CHAINS)); ghz.push_back((double)h[i] / (double)h[i + 1] * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double c = cyc[cyc.size() / 2], g = ghz[ghz.size() / 2];
    // by WALL time (steady state, see above): every SIMD ran wps waves x CH x CHAINS hashes per launch
    const double wall_cpi = ms * 1e-3 * g * 1e9 / ((double)wps * CH * CHAINS * kValuPerInner);
    printf("%d chain(s) per lane, %d wave(s) per SIMD: one wave's lifetime %7.0f cycles per hash = %.2f us at %.2f GHz (%.2f cycles per VALU instruction); "
           "steady-state wall time %.2f cycles = %.3f ns per VALU instruction per SIMD (%.1f us per launch)\n",
           CHAINS, wps, c, c / g / 1e3, g, c / kValuPerInner, wall_cpi, wall_cpi / g, ms * 1e3);
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
    return 0;
}
