// sha_quad_probe.hip -- the four-lane SHA-256 of sha256_quad.hpp against the one-lane sha256_inner: every wave hashes 16
// messages both ways and counts mismatching digest words; then a dependent chain of CH hashes through LDS (the shape of a
// latency-bound tree level) is timed on waves that have their SIMD to themselves.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zkstark_amd/csrc -o tools/sha_quad_probe tools/sha_quad_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "sha256_quad.hpp"

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace zk;
constexpr int CH = 64;

// mode 0: quad; mode 1: one lane per hash (lanes 0-15 of each wave active, same 16 messages)
template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* out, uint32_t seed, unsigned long long* rec, uint32_t* bad) {
    __shared__ __attribute__((aligned(16))) uint32_t msg[4][16][16];      // [wave][slot][word]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t slot = (lane >> 4) * 4 + (lane & 3u), role = (lane >> 2) & 3u;
    const QuadLane q = quad_lane(lane);
    if (lane < 16)
        for (int i = 0; i < 16; ++i) msg[wave][lane][i] = seed * (i + 1) + (blockIdx.x * 4 + wave) * 977u + lane * 131u;
    __builtin_amdgcn_wave_barrier();
    // correctness: slot's message both ways
    if (MODE == 0) {
        uint32_t o[4];
        sha256_inner_quad(msg[wave][slot], q, o);
        Digest l, r;
        for (int i = 0; i < 8; ++i) { l.w[i] = msg[wave][slot][i]; r.w[i] = msg[wave][slot][8 + i]; }
        Digest d = sha256_inner(l, r);
        uint32_t miss = 0;
        if (role == 1) for (int i = 0; i < 4; ++i) miss += o[i] != d.w[i];
        if (role == 0) for (int i = 0; i < 4; ++i) miss += o[i] != d.w[4 + i];
        if (miss) atomicAdd(bad, miss);
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = 0;
#pragma unroll 1
    for (int it = 0; it < CH; ++it) {
        if (MODE == 0) {
            uint32_t o[4];
            sha256_inner_quad(msg[wave][slot], q, o);
            __builtin_amdgcn_wave_barrier();
            if (role < 2) {                                              // next message: (digest, digest ^ seed)
                uint32_t* m = msg[wave][slot] + (role == 1 ? 0 : 4);
                for (int i = 0; i < 4; ++i) { m[i] = o[i]; m[8 + i] = o[i] ^ seed; }
            }
            x ^= o[0];
            __builtin_amdgcn_wave_barrier();
        } else {
            if (lane < 16) {
                Digest l, r;
                for (int i = 0; i < 8; ++i) { l.w[i] = msg[wave][lane][i]; r.w[i] = msg[wave][lane][8 + i]; }
                Digest d = sha256_inner(l, r);
                for (int i = 0; i < 8; ++i) { msg[wave][lane][i] = d.w[i]; msg[wave][lane][8 + i] = d.w[i] ^ seed; }
                x ^= d.w[0];
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (lane < 16 && MODE == 1) out[blockIdx.x * 256 + threadIdx.x] = msg[wave][lane][0];
    if (MODE == 0 && role == 1 && (lane & 3u) == 0 && lane < 16) out[blockIdx.x * 256 + threadIdx.x] = msg[wave][slot][0];
    if (lane == 0) {
        rec[((size_t)blockIdx.x * 4 + wave) * 2] = c1 - c0;
        rec[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
    }
}

template <int MODE>
int run(int cus, uint32_t* d_out, unsigned long long* d_rec, uint32_t* d_bad, uint32_t* first_word) {
    const int blocks = cus;                       // 4 waves per workgroup, one workgroup per CU: one wave per SIMD
    CHK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 3u, d_rec, d_bad);
    CHK(hipDeviceSynchronize());
    uint32_t bad = 0;
    CHK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> h((size_t)blocks * 8);
    CHK(hipMemcpy(h.data(), d_rec, h.size() * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(first_word, d_out + (MODE == 0 ? 4 : 0), 4, hipMemcpyDeviceToHost));   // slot 0 of wave 0: bank-1 lane 4 / lane 0
    std::vector<double> cyc, ghz;
    for (size_t i = 0; i < h.size(); i += 2) { cyc.push_back((double)h[i] / CH); ghz.push_back((double)h[i] / (double)h[i + 1] * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double c = cyc[cyc.size() / 2], g = ghz[ghz.size() / 2];
    printf("%s: %7.0f cycles per dependent hash = %.2f us at %.2f GHz; mismatching digest words %u; chain end %08x\n",
           MODE == 0 ? "four lanes per hash (sha256_inner_quad)" : "one lane per hash (sha256_inner)      ", c, c / g / 1e3, g, bad, *first_word);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *d_out, *d_bad; unsigned long long* d_rec;
    CHK(hipMalloc(&d_out, (size_t)cus * 256 * 4));
    CHK(hipMalloc(&d_bad, 4));
    CHK(hipMalloc(&d_rec, (size_t)cus * 4 * 16));
    uint32_t w0 = 0, w1 = 0;
    if (run<0>(cus, d_out, d_rec, d_bad, &w0)) return 1;
    if (run<1>(cus, d_out, d_rec, d_bad, &w1)) return 1;
    printf("chains agree: %s\n", w0 == w1 ? "yes" : "NO");
    return 0;
}
