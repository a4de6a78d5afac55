// valu_mix_probe2.hip -- why does SHA-256 (36 % "fast" VALU ops) issue at 4.05 cycles per instruction when an
// independent S S F stream issues at 3.3?  Candidates: dependent chains, SGPR / literal operands, SALU in the stream.
// Build: hipcc -O3 --offload-arch=gfx950 -I zkstark_amd/csrc -o tools/valu_mix_probe2 tools/valu_mix_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "sha256.hpp"
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 256, UNROLL = 16;

#define S2(d, x, y) asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(d) : "v"(x), "v"(y))
#define A3(d, x, y, z) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z))
#define A3S(d, x, y, z) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "s"(z))
#define F3(d, x, y, z) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(d) : "v"(x), "v"(y), "v"(z))
#define FADD(d, x, y) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define FADDL(d, x) asm volatile("v_add_u32 %0, 0x428a2f98, %1" : "=v"(d) : "v"(x))

// PAT 0: one dependent chain  S S F (each instruction reads the previous result)
// PAT 1: two interleaved dependent chains
// PAT 2: SHA-like round skeleton on real dependencies: 3 S (rot of e) -> F(xor3) -> F(ch) -> A3 -> A3 ; 3 S (rot of a) -> F -> F(maj) -> A3 ; FADD
// PAT 3: PAT 2 with the K constant from an SGPR and an s_mov per round (as the compiler emits)
// PAT 4: independent S S F with an SGPR operand on every add3-class op and a literal on the fast add
template <int PAT>
__global__ __launch_bounds__(256) void bench(uint32_t* out, uint32_t seed, uint32_t kc) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 5 + 1, d = seed * 7 + 2, e = seed * 11 + threadIdx.x, f = seed * 13, g = seed * 17, h = seed * 19;
    uint32_t t0, t1, t2, t3, t4;
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (PAT == 0) { S2(a, a, b); S2(a, a, c); F3(a, a, b, c); S2(a, a, d); S2(a, a, b); F3(a, a, c, d); S2(a, a, b); S2(a, a, c); F3(a, a, b, d); }
            if (PAT == 1) { S2(a, a, b); S2(e, e, f); S2(a, a, c); S2(e, e, g); F3(a, a, b, c); F3(e, e, f, g); S2(a, a, d); S2(e, e, h); S2(a, a, b); S2(e, e, f); F3(a, a, c, d); F3(e, e, g, h);
                            S2(a, a, b); S2(e, e, f); S2(a, a, c); S2(e, e, g); F3(a, a, b, d); F3(e, e, f, h); }
            if (PAT == 2 || PAT == 3) {
                S2(t0, e, e); S2(t1, e, e); S2(t2, e, e); F3(t0, t0, t1, t2); F3(t1, e, f, g);
                A3(h, h, t0, t1);
                if (PAT == 2) { A3(h, h, c, d); } else { uint32_t kk; asm volatile("s_mov_b32 %0, 0x428a2f98" : "=s"(kk)); A3S(h, h, c, kk); }
                S2(t2, a, a); S2(t3, a, a); S2(t4, a, a); F3(t2, t2, t3, t4); F3(t3, a, b, c);
                FADD(d, d, h);
                A3(h, h, t2, t3);
                // rotate the state names: (a..h) <- (h, a, b, c, d, e, f, g)
                uint32_t nh = g; g = f; f = e; e = d; d = c; c = b; b = a; a = h; h = nh;
            }
            if (PAT == 4) { S2(a, a, b); A3S(c, c, d, kc); FADDL(e, e); S2(f, f, g); A3S(h, h, b, kc); FADDL(a, a); S2(c, c, d); A3S(e, e, f, kc); FADDL(g, g); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

// PAT 5: the real sha256_inner, CH hashes in a chain (as tools/sha_latency_probe.hip), by wall time
__global__ __launch_bounds__(256) void sha_bench(uint32_t* out, uint32_t seed) {
    zk::Digest d;
#pragma unroll
    for (int i = 0; i < 8; ++i) d.w[i] = seed * (i + 1) + threadIdx.x + blockIdx.x * 977;
#pragma unroll 1
    for (int it = 0; it < 16; ++it) { zk::Digest r = d; r.w[0] ^= seed; d = zk::sha256_inner(d, r); }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= d.w[i];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int PAT>
int run(const char* name, int per_iter, uint32_t* d_out, int cus) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps, reps = 10;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        auto launch = [&](uint32_t sd) {
            if (PAT == 5) hipLaunchKernelGGL(sha_bench, dim3(blocks), dim3(256), 0, 0, d_out, sd);
            else hipLaunchKernelGGL(bench<(PAT == 5 ? 0 : PAT)>, dim3(blocks), dim3(256), 0, 0, d_out, sd, 0x71374491u);
        };
        launch(1);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch(12345u + r);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = PAT == 5 ? (double)reps * wps * 16 * 2293 : (double)reps * wps * ITER * UNROLL * per_iter;
        printf("%-40s %d waves/SIMD: %6.3f ns per VALU instruction per SIMD (%.2f cycles at 2.1 GHz)\n", name, wps, ms * 1e6 / instr, ms * 1e6 / instr * 2.1);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
    const int cus = prop.multiProcessorCount;
    run<0>("one dependent chain S S F", 9, d_out, cus);
    run<1>("two dependent chains S S F", 18, d_out, cus);
    run<2>("SHA round skeleton (VGPR operands)", 14, d_out, cus);
    run<3>("SHA round skeleton (K in SGPR + s_mov)", 14, d_out, cus);
    run<4>("independent S A3s Fl (SGPR, literal)", 9, d_out, cus);
    run<5>("sha256_inner chain (compiled code)", 0, d_out, cus);
    return 0;
}
