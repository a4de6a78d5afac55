// valu_mix_probe.hip -- does the 2-cycle issue of the "fast" VALU ops (v_bitop3_b32, v_add_u32, v_xor_b32: 110 lanes/clk/CU
// alone, tools/valu_microbench.hip) survive in a MIXED stream with 4-cycle ops (v_alignbit_b32, v_add3_u32), and does the
// order matter?  SHA-256's round is 9 slow + 5 fast ops; if fast ops only pair with adjacent fast ops, an instruction
// order that groups them would be worth ~20 %.  Patterns (S = v_alignbit_b32, F = v_bitop3_b32), all on independent chains:
//   0: SSSS SSSS (all slow)   1: FFFF FFFF (all fast)   2: SFSF SFSF   3: SSFF SSFF   4: SSSS FFFF   5: SSF SSF SSF (2:1)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/valu_mix_probe tools/valu_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 256, UNROLL = 16, ACC = 8;

#define SLOW(i) asm volatile("v_alignbit_b32 %0, %1, %2, 7" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]))
#define FAST(i) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "v"(b[(i + 1) & 7]))

template <int PAT>
__global__ __launch_bounds__(256) void bench(uint32_t* out, uint32_t seed) {
    uint32_t a[ACC], b[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) { a[i] = seed * (i + 1) + threadIdx.x; b[i] = (seed ^ 0x9e3779b9u) * (i + 3) + threadIdx.x * 7; }
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (PAT == 0) { SLOW(0); SLOW(1); SLOW(2); SLOW(3); SLOW(4); SLOW(5); SLOW(6); SLOW(7); }
            if (PAT == 1) { FAST(0); FAST(1); FAST(2); FAST(3); FAST(4); FAST(5); FAST(6); FAST(7); }
            if (PAT == 2) { SLOW(0); FAST(1); SLOW(2); FAST(3); SLOW(4); FAST(5); SLOW(6); FAST(7); }
            if (PAT == 3) { SLOW(0); SLOW(1); FAST(2); FAST(3); SLOW(4); SLOW(5); FAST(6); FAST(7); }
            if (PAT == 4) { SLOW(0); SLOW(1); SLOW(2); SLOW(3); FAST(4); FAST(5); FAST(6); FAST(7); }
            if (PAT == 5) { SLOW(0); SLOW(1); FAST(2); SLOW(3); SLOW(4); FAST(5); SLOW(6); SLOW(7); FAST(0); }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int PAT>
int run(const char* name, int per_iter, int n_fast, uint32_t* d_out, int cus) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(bench<PAT>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
        CHK(hipDeviceSynchronize());
        const int reps = 10;
        CHK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(bench<PAT>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)reps * wps * ITER * UNROLL * per_iter;
        const double ns_per_instr = ms * 1e6 / instr_per_simd;
        printf("%-22s (%d fast of %d) %d waves/SIMD: %6.3f ns per instruction per SIMD = %.2f cycles at 2.1 GHz; if fast = 2 and slow = 4 cycles: %.2f\n", name, n_fast, per_iter,
               wps, ns_per_instr, ns_per_instr * 2.1, (4.0 * (per_iter - n_fast) + 2.0 * n_fast) / per_iter);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
    run<0>("SSSS SSSS", 8, 0, d_out, prop.multiProcessorCount);
    run<1>("FFFF FFFF", 8, 8, d_out, prop.multiProcessorCount);
    run<2>("SFSF SFSF", 8, 4, d_out, prop.multiProcessorCount);
    run<3>("SSFF SSFF", 8, 4, d_out, prop.multiProcessorCount);
    run<4>("SSSS FFFF", 8, 4, d_out, prop.multiProcessorCount);
    run<5>("SSF SSF SSF", 9, 3, d_out, prop.multiProcessorCount);
    return 0;
}
